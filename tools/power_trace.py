"""Board power and shader clock while a workload runs, from the amdgpu hwmon files (power1_input in uW, freq1_input in Hz; readable without privileges).

    python tools/power_trace.py OUT.json [workload ...]

The parent only reads sysfs (it never touches the GPU); each workload runs as a child process that prints `WINDOW t0 t1 us_per_launch` (time.time() around its steady
loop), and the samples inside that window are summarised.  Workloads: step (bench.py's step, forward + loss + backward at batch 160), nt / nt_zero (the persistent NT
GEMM at the gate-up shape on random / all-zero operands: a power-bound kernel runs faster on zeros, an issue-bound one does not), tn_group (a block's grouped weight
gradients), attn_fwd, attn_bwd, rows (RMSNorm forward, HBM-bound), mfma0 / mfma1 / mfma2 (tools/microbench/mfma_power.hip built to /tmp/mfma_power: the matrix pipe alone
on zero / random / random-through-LDS operands -- the rate the power cap allows, which is the ceiling the GEMMs are to be read against)."""
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(which, seconds):
    sys.path.insert(0, ROOT)
    import torch

    from llm_quest_amd import _lib as L
    from llm_quest_amd import kernels as K

    B = int(os.environ.get("POWER_BATCH", "160"))
    M = B * 709
    r = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
    z = lambda *s: torch.zeros(*s, device="cuda", dtype=torch.bfloat16)
    if which in ("nt", "nt_zero"):
        mk = z if which == "nt_zero" else r
        a, b = mk(M, 1024), mk(6144, 1024)
        out = K.gemm(L.GEMM_NT, a, b)
        run = lambda: K.gemm(L.GEMM_NT, a, b, out=out)
    elif which in ("tn_group", "tn_group_zero"):
        mk = z if which.endswith("zero") else r
        shapes = [(4096, 1024), (1024, 2048), (6144, 1024), (1024, 3072)]
        probs = [(mk(M, o), mk(M, i), torch.zeros(o, i, device="cuda", dtype=torch.bfloat16), None) for o, i in shapes]
        run = lambda: K.gemm_grouped(L.GEMM_TN, probs)
    elif which in ("attn_fwd", "attn_bwd"):
        S, Hq, Hkv, D = 709, 16, 8, 128
        q, k, v, do = r(M, Hq * D), r(M, Hkv * D), r(M, Hkv * D), r(M, Hq * D)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        km = torch.ones(B, S, dtype=torch.uint8, device="cuda")
        o, lse = K.attn_fwd(q, k, v, B, S, Hq, Hkv, D, key_mask=km, causal=True)
        if which == "attn_fwd":
            run = lambda: K.attn_fwd(q, k, v, B, S, Hq, Hkv, D, key_mask=km, causal=True)
        else:
            run = lambda: K.attn_bwd(q, k, v, o, do, lse, B, S, Hq, Hkv, D, dq, dk, dv, key_mask=km, causal=True)
    elif which == "rows":
        x, w = r(M, 1024), torch.ones(1024, device="cuda", dtype=torch.bfloat16)
        run = lambda: K.rmsnorm_fwd(x, w, 1e-6)
    else:
        raise SystemExit(f"unknown workload {which}")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t = time.time()
    n = 0
    while time.time() - t < 1.0:  # size the loop
        run()
        n += 1
        if n % 16 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    per = (time.time() - t) / n
    reps = max(16, int(seconds / per))
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    s.record()
    for _ in range(reps):
        run()
    e.record()
    torch.cuda.synchronize()
    t1 = time.time()
    print(f"WINDOW {t0:.6f} {t1:.6f} {s.elapsed_time(e) / reps * 1e3:.2f}", flush=True)


def hwmon():
    """The hwmon directory of the card the children run on: a box shows the hwmon files of cards that are not this job's, so run a short GEMM loop and take
    the card whose power follows it."""
    dirs = [os.path.dirname(p) for p in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")]
    if not dirs:
        raise SystemExit("no amdgpu hwmon with power1_input is visible")
    rd = lambda d: int(open(d + "/power1_input").read()) / 1e6
    idle = {d: rd(d) for d in dirs}
    p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", "nt", "4"], stdout=subprocess.PIPE, text=True, cwd=ROOT)
    peak = dict(idle)
    while p.poll() is None:
        for d in dirs:
            peak[d] = max(peak[d], rd(d))
        time.sleep(0.05)
    rise = {d: peak[d] - idle[d] for d in dirs}
    best = max(rise, key=rise.get)
    print("hwmon candidates (idle W, rise W):", {d.split("/")[4]: (round(idle[d]), round(rise[d])) for d in dirs}, flush=True)
    if rise[best] < 150:
        raise SystemExit("no visible hwmon follows this job's load: the card's telemetry is not exposed on this box")
    return best


def pct(v, q):
    v = sorted(v)
    return v[min(len(v) - 1, int(q * len(v)))]


def main():
    out, loads = sys.argv[1], sys.argv[2:] or ["step", "nt", "nt_zero", "tn_group", "attn_fwd", "attn_bwd", "rows"]
    h = hwmon()
    cap = int(open(h + "/power1_cap").read()) / 1e6
    samples, stop = [], threading.Event()

    def sampler():
        fp, ff = open(h + "/power1_input"), open(h + "/freq1_input")
        while not stop.is_set():
            fp.seek(0), ff.seek(0)
            try:
                samples.append((time.time(), int(fp.read()) / 1e6, int(ff.read()) / 1e6))
            except (OSError, ValueError):
                pass
            time.sleep(0.004)

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    res = {"power_cap_W": cap, "hwmon": h, "workloads": {}}
    for w in loads:
        if w == "step":
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "3", "--cpu-baseline", "off"]
            t_start = time.time()
            p = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
            t_end = time.time()
            line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
            dur = line["ms_per_step"] * line["steps"] / 1e3
            t0, t1, per = t_end - 1.0 - dur, t_end - 1.0, line["ms_per_step"] * 1e3  # the timed region ends just before the process does
            t0 = max(t0, t_start)
        else:
            if w.startswith("mfma"):  # tools/microbench/mfma_power.hip built to /tmp/mfma_power: mfma0 / mfma1 / mfma2
                cmd = ["/tmp/mfma_power", w[4:], "6"]
            elif w.startswith("exe="):  # any binary that prints a WINDOW line: exe=/tmp/stream_power,2048,5
                cmd = w[4:].split(",")
            else:
                cmd = [sys.executable, os.path.abspath(__file__), "--child", w, "6"]
            p = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
            wl = [l for l in p.stdout.splitlines() if l.startswith("WINDOW")]
            if not wl:
                res["workloads"][w] = {"error": (p.stderr or p.stdout)[-400:]}
                continue
            t0, t1, per = (float(x) for x in wl[-1].split()[1:])
        # skip the first 30 % of the window: the firmware's power average and the clock settle there
        win = [s for s in samples if t0 + 0.3 * (t1 - t0) <= s[0] <= t1]
        pw, fq = [s[1] for s in win], [s[2] for s in win]
        res["workloads"][w] = {
            "us_per_launch": per, "samples": len(win),
            "power_W": {"mean": sum(pw) / max(1, len(pw)), "p10": pct(pw, 0.1), "p50": pct(pw, 0.5), "p90": pct(pw, 0.9), "max": max(pw)} if pw else None,
            "sclk_MHz": {"mean": sum(fq) / max(1, len(fq)), "p10": pct(fq, 0.1), "p50": pct(fq, 0.5), "p90": pct(fq, 0.9)} if fq else None,
        }
        print(w, json.dumps(res["workloads"][w]), flush=True)
        time.sleep(3.0)  # let the board cool to the same starting point
    stop.set()
    th.join()
    json.dump(res, open(out, "w"), indent=1)
    with open(out.replace(".json", "_samples.csv"), "w") as f:
        f.write("t,power_W,sclk_MHz\n")
        t00 = samples[0][0] if samples else 0
        for s in samples[::5]:
            f.write(f"{s[0] - t00:.3f},{s[1]:.0f},{s[2]:.0f}\n")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2], float(sys.argv[3]))
    else:
        main()
