"""Development tool: writes the projection matrices and coordinate volumes of the volumetric benchmark case for
tools/brick_stats (frames, S, H, W as int32, then proj, then coord, float32)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseestimation_amd import synth  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
case = synth.volumetric_case(1234 + 2, frames, 4, with_volumes=False)
with open(sys.argv[2] if len(sys.argv) > 2 else "/tmp/brick_case.bin", "wb") as f:
    np.array([frames, 64, 96, 96], dtype=np.int32).tofile(f)
    case["proj"].astype(np.float32).tofile(f)
    case["coord_volumes"].astype(np.float32).tofile(f)
