#!/usr/bin/env python3
"""ipx_link_probe under the environment it is started with (GPU_MAX_HW_QUEUES, IPX_LANES ...)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import imageprocessor_amd as ipx
ctx = ipx.Context(device=0)
print(os.environ.get("GPU_MAX_HW_QUEUES"), ctx.link_probe(512 << 20, 512 << 20, 3), ctx.link_probe(64 * 8294400, 64 * 11600000, 3))
