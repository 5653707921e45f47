"""Per-workgroup wall-clock timeline of one NT GEMM launch (profiling build -DGEMM_TL=1 of the tile's part, tools/build_variant.sh tl gemm_p2 "-DGEMM_TL=1",
MI355_LIB_PATH=build_variants/libmi355vlm_tl.so): where a CU's time goes between the MFMAs of one tile and the MFMAs of the next.  GPU box only.
Stamps (100 MHz): 0 entry, 1 first K-tile landed, 2 main loop done, 3 barrier in front of the write-out, 4 write-out issued, 5 stores retired."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from llm_quest_amd import kernels as K, _lib as L
lib = L.load()
r = lambda *s: (0.1 * torch.randn(*s, device="cuda")).to(torch.bfloat16)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 113440
shapes = {"qkv": (4096, 1024), "dctx": (2048, 1024), "n1024k1024": (1024, 1024), "down": (1024, 3072)}
TILE = int(os.environ.get("TL_TILE", "2"))  # 7: the persistent kernel (stamps 0 = tile's main loop begins, 2 = ends, 4 = write-out issued; indexed by tile)
for name in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["qkv", "down"]):
    N, Kd = shapes[name]
    x, w = r(M, Kd), r(N, Kd)
    o = K.gemm(L.GEMM_NT, x, w, tile=TILE)
    for _ in range(40): K.gemm(L.GEMM_NT, x, w, out=o, tile=TILE)
    torch.cuda.synchronize()
    ntile = ((M + 255) // 256) * ((N + 255) // 256)
    n = min(ntile, 32768)
    buf = np.zeros((n, 8), dtype=np.uint64)
    rc = lib.mi355_debug_gemm_tl(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), n)
    assert rc == 0, rc
    t = buf[:, :6].astype(np.int64)
    if TILE == 7:
        order = np.argsort(t[:, 0])
        ml, wo = (t[:, 2] - t[:, 0]) / 100.0, (t[:, 4] - t[:, 2]) / 100.0
        span = (t[:, 4].max() - t[:, 0].min()) / 100.0
        # a workgroup's consecutive tiles are q, q + 256, ...: the gap between write-out end and the next main loop's begin
        G = min(ntile, 256)
        gaps = np.array([(t[q + G, 0] - t[q, 4]) / 100.0 for q in range(n - G)])
        mhz = (t[:, 3] - t[:, 1]) / np.maximum(ml, 1e-3)  # shader-clock cycles per microsecond over the main loop
        print(f"   shader clock during the main loops: mean {mhz.mean():.0f} MHz (p10 {np.percentile(mhz, 10):.0f}, p90 {np.percentile(mhz, 90):.0f}); main loop {((t[:, 3] - t[:, 1])).mean():.0f} cycles per tile, "
              f"MFMA issue {2 * 64 * (Kd // 64) * 16} cycles per SIMD at 16 per 16x16x32 and two waves")
        print(f"== {name} persistent: M {M} N {N} K {Kd} tiles {ntile} span {span:.1f} us; per tile: main loop {ml.mean():.2f} (p10 {np.percentile(ml, 10):.2f}, p90 {np.percentile(ml, 90):.2f}), "
              f"write-out {wo.mean():.2f} (p10 {np.percentile(wo, 10):.2f}, p90 {np.percentile(wo, 90):.2f}), between tiles {gaps.mean():.2f}")
        continue
    hw = buf[:, 6]
    xcc = (hw >> np.uint64(32)) & np.uint64(15)
    hwid = hw & np.uint64(0xffffffff)
    cu = (hwid >> np.uint64(8)) & np.uint64(15); sh = (hwid >> np.uint64(12)) & np.uint64(1); se = (hwid >> np.uint64(13)) & np.uint64(7)
    key = (xcc.astype(np.int64) << 8) | (se.astype(np.int64) << 5) | (sh.astype(np.int64) << 4) | cu.astype(np.int64)
    t0 = t[:, 0].min()
    span = (t[:, 5].max() - t0) / 100.0
    ph = np.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3], t[:, 5] - t[:, 4]], 1) / 100.0
    print(f"== {name}: M {M} N {N} K {Kd}  tiles {ntile}  distinct CUs {len(np.unique(key))}  launch span {span:.1f} us")
    print("   per tile, us (mean / p10 / p90):")
    for i, lab in enumerate(["prologue (entry -> first K-tile landed)", "main loop", "barrier before write-out", "write-out issue", "stores retire"]):
        print(f"     {lab:42s} {ph[:, i].mean():7.2f} / {np.percentile(ph[:, i], 10):7.2f} / {np.percentile(ph[:, i], 90):7.2f}")
    gaps, busy, per_cu = [], [], []
    for k in np.unique(key):
        idx = np.where(key == k)[0]
        idx = idx[np.argsort(t[idx, 0])]
        per_cu.append(len(idx))
        g = (t[idx[1:], 0] - t[idx[:-1], 5]) / 100.0
        gaps.extend(g.tolist())
        busy.append(ph[idx, 1].sum())
    gaps = np.array(gaps)
    print(f"   gap between a CU's workgroups (stores retired -> next entry): mean {gaps.mean():.2f} us, p10 {np.percentile(gaps, 10):.2f}, p90 {np.percentile(gaps, 90):.2f}")
    print(f"   tiles per CU: min {min(per_cu)} max {max(per_cu)}; main-loop time per CU: mean {np.mean(busy):.1f} us of {span:.1f} ({100 * np.mean(busy) / span:.0f} %)")
    tot = ph.sum(1).mean() + gaps.mean()
    print(f"   a tile's turn: {tot:.2f} us = main loop {ph[:, 1].mean():.2f} + everything else {tot - ph[:, 1].mean():.2f}")
