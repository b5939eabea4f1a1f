"""Shim package standing in for the closed Isaac Gym binary (test infrastructure only).

Isaac Gym Preview 4 is not installable here.  This package lets the
*reference's own Python* (motion_lib, torch_utils, the jit reward/reset/obs functions)
be imported from /root/reference inside this container so that golden vectors can be
generated from it (oracle/gen_golden.py).  Only `torch_utils` carries real math; the
gymapi/gymtorch/gymutil stand-ins are attribute sinks that are never executed.
Nothing in the product path imports this.
"""
