#!/usr/bin/env python3
"""tools/lab/enc_times.py — where the fused matcher's wave-time goes (a lab build with -DENC_DEBUG_TIMES: clocks around the phases of
match_chunk, summed over all wavefronts, per byte plane).  HIPBLOSC_LIB must point at that build.  Lab tooling."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "go-blosc_amd"))
import numpy as np, torch
import hipblosc as hb, bench
L = hb.lib(); assert L.hb_init() == 0
n = 1 << 30
d = bench.Dev(n, torch.device("cuda", 0))
d.src.copy_(torch.from_numpy(bench.synth_host("f32", n, 0)).view(torch.uint8))
d.compress(1, 4, hb.OPT_INDEX_TRAILER); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
L.hblab_enc_times(buf, 1)
d.compress(1, 4, hb.OPT_INDEX_TRAILER); torch.cuda.synchronize()
L.hblab_enc_times(buf, 0)
t = np.array(buf[:], dtype=np.float64).reshape(8, 16)
names = ["full", "gated", "miss"]
for j in range(4):
    r = t[j]; chunks = max(r[13], 1)
    print(f"plane {j}: chunks {int(chunks)}  per chunk: total {r[12]/chunks:8.0f} clk | staging {r[11]/chunks:7.0f} | step loop {r[6]/chunks:8.0f} | tail {r[10]/chunks:7.0f}")
    print(f"          full steps {r[1]/chunks:5.1f} x {r[0]/max(r[1],1):6.0f} clk | gated {r[3]/chunks:5.1f} x {r[2]/max(r[3],1):6.0f} | miss {r[5]/chunks:5.1f} x {r[4]/max(r[5],1):6.0f} | flushes {r[9]/chunks:4.1f} x {r[8]/max(r[9],1):6.0f}")
