#!/usr/bin/env python
"""Report of a rocprofv3 --kernel-trace database taken over tools/step_trace.py: the windows between consecutive `spin_kernel` marks are
train steps.  Prints (1) per kernel: launches per step, average duration, time per step, share; (2) per step: span, union of the kernel
intervals (GPU busy), sum of the kernel durations (> span where launches overlap); (3) with --layers: every dispatch of the kernels that
run the discriminators' mid layers (tap3 / tap4 / bl_dw) of ONE step in dispatch order per hardware queue, so that a launch can be named
by its position in its chain.  Usage: step_trace_report.py p_results.db [--layers] [--top N]"""
import argparse, re, sqlite3
ap = argparse.ArgumentParser(); ap.add_argument("db"); ap.add_argument("--layers", action="store_true"); ap.add_argument("--top", type=int, default=70)
ap.add_argument("--main", action="store_true", help="every dispatch on the main stream's queue of one step, with the idle time of that queue before it")
ap.add_argument("--window", help="a,b: every dispatch of one step that starts between a and b ms after the step mark, all queues")
ap.add_argument("--named", help="write the per-launch in-step durations of MelGAN's MFMA-bound layers (named by their position in the chain) to this JSON file")
a = ap.parse_args()
con = sqlite3.connect(a.db)
suf = [r[0] for r in con.execute("select name from sqlite_master where type='table'") if r[0].startswith("rocpd_metadata")][0][len("rocpd_metadata"):]
cols = [r[1] for r in con.execute(f"pragma table_info(rocpd_kernel_dispatch{suf})")]
qcol = "queue_id" if "queue_id" in cols else "stream_id"
gcol = "grid_size_x" if "grid_size_x" in cols else ("grid_x" if "grid_x" in cols else None)
sel = f"select s.kernel_name, d.start, d.end, d.{qcol}" + (f", d.{gcol}" if gcol else ", 0") + f" from rocpd_kernel_dispatch{suf} d join rocpd_info_kernel_symbol{suf} s on d.kernel_id = s.id order by d.start"
rows = con.execute(sel).fetchall()
def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("eben::", "")
    return n[:84]
marks = [s for n, s, e, q, g in rows if "spin_kernel" in n]
assert len(marks) >= 2, "no step marks (run tools/step_trace.py under rocprofv3 --kernel-trace)"
nsteps = len(marks) - 1
t0, t1 = marks[0], marks[-1]
agg = {}
for n, s, e, q, g in rows:
    if s < t0 or s >= t1 or "spin_kernel" in n: continue
    k = short(n)
    v = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
    d = (e - s) / 1e3
    v[0] += 1; v[1] += d; v[2] = min(v[2], d); v[3] = max(v[3], d)
tot = sum(v[1] for v in agg.values())
print(f"# {nsteps} steps, {(t1 - t0) / 1e6 / nsteps:.3f} ms per step (mark to mark), {tot / 1e3 / nsteps:.3f} ms of kernel time per step, {sum(v[0] for v in agg.values()) / nsteps:.0f} dispatches per step")
print(f"{'per step':>9} {'avg us':>9} {'min us':>9} {'ms/step':>9} {'pct':>6}  kernel")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[: a.top]:
    print(f"{v[0] / nsteps:9.1f} {v[1] / v[0]:9.1f} {v[2]:9.1f} {v[1] / 1e3 / nsteps:9.3f} {100 * v[1] / tot:6.2f}  {k}")
print("\n# per step: span ms, GPU busy ms (union of kernel intervals), idle ms")
for i in range(nsteps):
    iv = sorted((s, e) for n, s, e, q, g in rows if marks[i] <= s < marks[i + 1] and "spin_kernel" not in n)
    busy, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None: busy += cur_e - cur_s
    span = marks[i + 1] - marks[i]
    print(f"step {i}: span {span / 1e6:.3f}  busy {busy / 1e6:.3f}  idle {(span - busy) / 1e6:.3f}")
if a.layers:
    i = nsteps // 2
    qs = {}
    print(f"\n# step {i}: dispatches of the discriminator mid-layer kernels in dispatch order (ms from the step mark, us, queue, grid, kernel)")
    for n, s, e, q, g in rows:
        if not (marks[i] <= s < marks[i + 1]): continue
        if not any(t in n for t in ("tap3_kernel", "tap4_kernel", "bl_dw_kernel", "bl_head", "bl_tail", "bl_fm")): continue
        qs.setdefault(q, len(qs))
        print(f"{(s - marks[i]) / 1e6:8.3f} {(e - s) / 1e3:8.1f}  q{qs[q]}  grid {g:>8}  {short(n)}")

if a.main:
    i = nsteps // 2
    mq = next(q for n, s, e, q, g in rows if "spin_kernel" in n)
    print(f"\n# step {i}: the main stream's queue (ms from the step mark, us, idle us before, kernel); other queues' dispatches counted only")
    last = marks[i]
    idle_tot = busy_tot = 0.0
    small = {}
    for n, s, e, q, g in rows:
        if not (marks[i] <= s < marks[i + 1]) or q != mq or "spin_kernel" in n: continue
        gap = max(0.0, (s - last) / 1e3)
        idle_tot += gap; busy_tot += (e - s) / 1e3
        print(f"{(s - marks[i]) / 1e6:8.3f} {(e - s) / 1e3:8.1f} {gap:8.1f}  {short(n)}")
        last = max(last, e)
    print(f"# main queue: busy {busy_tot / 1e3:.3f} ms, idle {idle_tot / 1e3:.3f} ms")

if a.window:
    w0, w1 = (float(v) for v in a.window.split(","))
    i = nsteps // 2
    qs = {}
    print(f"\n# step {i}: dispatches starting {w0} .. {w1} ms after the step mark (ms, us, queue, grid, kernel)")
    for n, s, e, q, g in rows:
        qs.setdefault(q, len(qs))
        if not (marks[i] + w0 * 1e6 <= s < marks[i] + w1 * 1e6) or "spin_kernel" in n: continue
        print(f"{(s - marks[i]) / 1e6:8.3f} {(e - s) / 1e3:8.1f}  q{qs[q]}  grid {g:>8}  {short(n)}")

if a.named:
    # MelGAN's chain runs on the queue of its head (bl_head_fwd_kernel<1, 16, 15>).  On that queue the persistent tile kernel tap4<2,2,4,...>
    # runs layers 3, 4, 5 forward right before the logits layer (bl_tail_fwd) and layers 5, 4, 3 of the input gradients right behind its
    # gradient (bl_tail_dx) -- once per pass of the two-pass backward.  Every such launch inside the marked steps is listed.
    import json
    mq = next((q for n, s, e, q, g in rows if "bl_head_fwd_kernel" in n and ("ILi1ELi16ELi15E" in n or "<1, 16, 15>" in n)), None)
    assert mq is not None, "no MelGAN head launch in the trace"
    seq = [(n, s, e) for n, s, e, q, g in rows if q == mq and t0 <= s < t1]
    is_t4 = lambda n: "tap4_kernel" in n and ("ILi2ELi2ELi4E" in n or "<2, 2, 4," in n)
    named = {}
    for i, (n, s, e) in enumerate(seq):
        if "bl_tail_fwd" in n:
            back = [x for x in seq[max(0, i - 8):i] if is_t4(x[0])][-3:]
            if len(back) == 3:
                for name, x in zip(("melgan_l3_fwd", "melgan_l4_fwd", "melgan_l5_fwd"), back):
                    named.setdefault(name, []).append((x[2] - x[1]) / 1e3)
        if "bl_tail_dx" in n:
            fwd = [x for x in seq[i + 1:i + 9] if is_t4(x[0])][:3]
            if len(fwd) == 3:
                for name, x in zip(("melgan_l5_dx", "melgan_l4_dx", "melgan_l3_dx"), fwd):
                    named.setdefault(name, []).append((x[2] - x[1]) / 1e3)
    out = {k: {"launches": len(v), "avg_us": round(sum(v) / len(v), 2), "min_us": round(min(v), 2), "max_us": round(max(v), 2)} for k, v in named.items()}
    out["_note"] = (f"rocprofv3 --kernel-trace of tools/step_trace.py, {nsteps} marked steps of the graph-replayed train step ({(t1 - t0) / 1e6 / nsteps:.2f} ms per step under "
                    "the tracer); launches named by their position on MelGAN's queue; the input-gradient launches are 2B-row passes (rows [fm | adv], then "
                    "[fake | real])")
    json.dump(out, open(a.named, "w"), indent=1)
    print("\n# named launches (us): " + ", ".join(f"{k} {v['avg_us']} (min {v['min_us']}, {v['launches']}x)" for k, v in out.items() if not k.startswith("_")))
