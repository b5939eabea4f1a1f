"""vid2player3d_amd — MI355X-native rollout engine behind vid2player3d's embodied_pose VecTask surface.

Only the hot path of SURVEY.md §8 lives here: HIP kernels + C-ABI (`csrc/`), the host-side
mirror of the reference task/motion-lib interface, and env sharding helpers.
"""
__version__ = "0.1.0"
