"""Long-run sanity of the engine on the GPU box: python tools/soak.py [steps] [racket|humanoid] [num_envs] [pgs|tgs] [world|velocity]  (racket: the racket + ball task, a
ball served at every player each epoch: ball x hull / racket / ground contacts, joint limits, substep jobs; num_envs <= 5120 runs the
library's register build, 6000 envs cut into substep jobs there)."""
import sys, torch
sys.path.insert(0, ".")
import bench
RACKET = len(sys.argv) > 2 and sys.argv[2] == "racket"
NENV = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
SOLVER = sys.argv[4] if len(sys.argv) > 4 else "pgs"
FRAME = sys.argv[5] if len(sys.argv) > 5 else "world"
task = bench.build_task(NENV, 0, 7, djokovic=RACKET, racket_ball=RACKET, substep_jobs=True, solver=SOLVER, env_extra={"friction_frame": FRAME})
g = torch.Generator(device="cuda"); g.manual_seed(1)
n = task.num_envs
bad = 0
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2000):
    if i % 32 == 0:
        task.reset()
    a = bench.make_actions(task, 0.25 * torch.randn((n, 75), device="cuda", generator=g))
    if i % 7 == 3:   # staged path now and then (exercises the pairing fallbacks)
        task.pre_physics_step(a); task._physics_step(); task.post_physics_step()
    elif i % 11 == 5:
        task._physics_step(); task._physics_step()
        task.step(a)
    else:
        task.step(a)
    if i % 100 == 99:
        torch.cuda.synchronize()
        ok = bool(torch.isfinite(task.obs_buf).all() and torch.isfinite(task._rigid_body_state).all() and torch.isfinite(task.rew_buf).all())
        perm, key = task.debug_pairing()
        okp = bool((torch.sort(perm.long()).values == torch.arange(n, device="cuda")).all())
        zmin = float(task._rigid_body_state.view(n, 24, 13)[..., 2].min())
        extra = ""
        if RACKET:
            bs = task._ball_root_states
            ok = ok and bool(torch.isfinite(bs).all()) and float(bs[:, 2].min()) > 0.0 and float(bs[:, 7:10].norm(dim=1).max()) < 120.0
            extra = " ball z %.3f..%.2f |v|max %.1f bounced %.2f hit %.3f" % (float(bs[:, 2].min()), float(bs[:, 2].max()), float(bs[:, 7:10].norm(dim=1).max()),
                                                                            float(task._has_bounce.float().mean()), float(task._has_racket_ball_contact.float().mean()))
        print(i + 1, "finite", ok, "perm ok", okp, "zmin %.3f" % zmin, "alive %.3f" % float((task.reset_buf == 0).float().mean()) + extra)
        bad += (not ok) + (not okp)
task.check()
print("substep jobs recomputed after waiting in vain:", task.job_recoveries())
print("SOAK", "OK" if bad == 0 else "FAILED", "| envs", n, "| build:", task.kernel_build(), "| solver", task.contact_solver, "| friction frame", task.friction_frame)
