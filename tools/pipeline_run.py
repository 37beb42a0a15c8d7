"""Renders `frames` frames of rank 0's share (tile_step G) with K frames in flight and nothing else, for tracing:
python tools/pipeline_run.py c2 8 4 40"""
import os, sys
import os as _os; _os.environ.setdefault("HIPEXEC_KNOBS", "1")   # the HIPEXEC_* experiment knobs are honoured only with this set
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import scenegen
from raytracer_amd.executor import FramesInFlight
wl, G, K, frames = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
W, H, depth, fn = {"c2": (1920, 1080, 3, scenegen.scene_c2), "c3": (3840, 2160, 5, scenegen.scene_c2)}[wl]
fs = fn().to_flat()
pipe = FramesInFlight(fs, device=0, depth=K)
outs = [torch.zeros((((H + 7) // 8) * 8, W, 4), dtype=torch.float32, device="cuda") for _ in range(K)]
for _ in range(frames):
    k = pipe.next
    pipe.submit(W, H, outs[k].data_ptr(), None, max_depth=depth, tile_first=0, tile_step=G)
pipe.drain()
pipe.close()
