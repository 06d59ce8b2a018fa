import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, lsm_amd as lsm
from bench import build_equation, one_step
eq, grid, vel = build_equation(lsm, (512, 512, 512), None, 0, "fast")
tc = 0.0
for _ in range(3): tc = one_step(eq, tc)
for prof in (False, True, False, True):
    eq.backend.profile_enable(prof)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): tc = one_step(eq, tc)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    r = eq.backend.profile_read() if prof else (0, 0.0)
    print("profile events", prof, "ms/step %.4f" % (el / 20 * 1e3), "stage ms/launch %.4f" % (r[1] / max(1, r[0])))
