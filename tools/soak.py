import sys, torch
sys.path.insert(0, ".")
import bench
task = bench.build_task(8192, 0, 7)
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
        print(i + 1, "finite", ok, "perm ok", okp, "zmin %.3f" % zmin, "alive %.3f" % float((task.reset_buf == 0).float().mean()))
        bad += (not ok) + (not okp)
print("SOAK", "OK" if bad == 0 else "FAILED")
