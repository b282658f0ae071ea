"""Quick GPU check of the lane kernel: C3 placements canvas through CompositeBatch vs the oracle; time lane vs march."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle")); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import numpy as np, torch
import oracle
from image_transformation_amd import synthetic, _native
from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, coerce_placements
for amode in ("soft", "binary"):
    size, objs, pl = synthetic.placements_workload(3840, 2160, 32, 3, amode)
    a = Atlas(objs)
    plan = CompositeBatch(a, [SolidCanvas(size, synthetic.SOLID_BG)], [coerce_placements(a, pl)])
    out = plan.run()[0]
    torch.cuda.synchronize()
    print(amode, plan.stats())
    bg = np.empty((size[1], size[0], 4), np.uint8); bg[:] = synthetic.SOLID_BG
    want = oracle.composite(bg, objs, pl)
    got = out.cpu().numpy()
    bad = int((got != want).any(axis=2).sum())
    print(amode, "mismatching pixels:", bad, "of", got.shape[0] * got.shape[1])
    ctx = a.ctx
    outs = [plan.alloc_outputs()]
    for k in range(3):
        plan.invalidate(); plan.run(outs[0])
    ctx.profile_begin(20)
    for k in range(20):
        plan.invalidate(); plan.run(outs[0], check=False)
    torch.cuda.synchronize()
    calls, c_ms, r_ms = ctx.profile_end()
    print(f"{amode}: resample {r_ms / calls * 1e3:.2f} us, composite {c_ms / calls * 1e3:.2f} us (MIC_RS_LANE={os.environ.get('MIC_RS_LANE', '1')})")
