import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import imageprocessor_amd as ipx, oracle
from helpers import rgba_frames
ctx = ipx.Context(lanes=2)
for (sw, sh, rs) in ((640, 480, (1024, 768, True)), (320, 240, (512, 384, False)), (256, 64, (300, 100, False))):
    frames = rgba_frames(2, sw, sh, seed=5)
    plan = ctx.plan(sw, sh, resize=rs, thumbnail=None, watermark=None)
    got = plan.run_host(frames)["resize"]
    want = oracle.process(frames[0], resize=rs, want=("resize",))["resize"]
    bad = np.argwhere((got[0] != want).any(-1))
    print(sw, sh, rs, "mismatched px", len(bad), "of", want.shape[0] * want.shape[1])
    if len(bad):
        ys = np.unique(bad[:, 0]); xs = np.unique(bad[:, 1])
        print(" rows", ys[:10], "..", ys[-5:], len(ys), " cols", xs[:10], "..", xs[-5:], len(xs))
        y, x = bad[0]; print(" first", y, x, got[0][y, x], want[y, x])
    plan.close()
