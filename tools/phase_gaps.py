"""Where does wall - kernel time of the default train step sit?  (verdict r5 item 5: attribute the gaps OUTSIDE the profiler.)

    python tools/phase_gaps.py events            -> per-phase GPU spans of the un-profiled step (HIP events on the launch stream at the phase
                                                    boundaries: augment | forward | losses | backward | optimizer), JSON on stdout
    python tools/phase_gaps.py trace <kernel_trace.csv> [events.json]
                                                 -> per-phase launches, kernel time and span from a rocprofv3 kernel trace of the same loop
                                                    (phases cut at kernel names), next to the un-profiled spans: span - kernel time = what the
                                                    boundaries between dependent launches cost in that phase
Same workload as `python bench.py` (configs[2]).
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PHASES = ["augment", "forward", "losses", "backward", "optimizer"]


def run_events(steps=12, warmup=5):
    import torch
    from scl_amd import augment
    from scl_amd.datautils_common import default_rawboost_args
    from scl_amd.model_linear import Model
    from scl_amd.optim import FusedAdamW
    dev = torch.device("cuda:0")
    model = Model({"flag_fix_ssl": False, "contra_mode": "all", "loss_type": 1}, dev, seed=0)
    model.train()
    opt = FusedAdamW(model, lr=1e-5, weight_decay=1e-4)
    conf = {"model": {"contra_mode": "all", "loss_type": 1}}
    B, L = 64, 64000
    x = (0.1 * torch.randn(B, L, generator=torch.Generator().manual_seed(1234))).to(dev)
    y = torch.tensor(([1] * ((5 * B + 10) // 11) + [0] * B)[:B], device=dev)
    rb = default_rawboost_args()
    ev = lambda: torch.cuda.Event(enable_timing=True)
    marks = []

    def step(rec):
        e = [ev() for _ in range(6)] if rec else None
        if rec: e[0].record()
        xs = augment.rawboost_batch(x, rb, 5, 16000, sampler="fast")
        if rec: e[1].record()
        out, feats, emb = model(xs)
        if rec: e[2].record()
        losses = model.loss(out, feats, emb, y, conf)
        total = None
        for v in losses.values():
            total = v if total is None else total + v
        if rec: e[3].record()
        opt.zero_grad()
        total.backward()
        if rec: e[4].record()
        opt.step()
        if rec:
            e[5].record()
            marks.append(e)
    for _ in range(warmup):
        step(False)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(steps):
        step(os.environ.get("PHASE_NO_EVENTS") != "1")
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    res = {"ms_per_step_wall": wall, "steps": steps}
    if marks:
        spans = [[m[i].elapsed_time(m[i + 1]) for i in range(5)] for m in marks]
        # the loss events bracket the FORWARD part of the losses; their backward runs inside total.backward() — reported with "backward".
        # The FIRST step after the synchronize is kept apart: the GPU is idle while the launch thread draws and designs that step's
        # RawBoost parameters (in every later step that host work runs while the GPU is still busy with the step before).
        res["first_step_spans_ms"] = {PHASES[i]: spans[0][i] for i in range(5)}
        rest = spans[1:]
        res["spans_ms"] = {PHASES[i]: sum(s[i] for s in rest) / len(rest) for i in range(5)}
        res["step_span_ms"] = sum(m[0].elapsed_time(m[5]) for m in marks[1:]) / len(rest)
        res["between_steps_ms"] = sum(marks[i][5].elapsed_time(marks[i + 1][0]) for i in range(len(marks) - 1)) / max(1, len(marks) - 1)
    print(json.dumps(res))


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def run_trace(path, events_json=None, n=3):
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(path)))
    idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
    agg = collections.OrderedDict((p, {"n": 0, "k": 0.0, "span": 0.0}) for p in PHASES)
    gaps_after = collections.defaultdict(lambda: [0, 0.0])
    for st in range(n):
        lo, hi = idx[-2 - st] + 1, idx[-1 - st] + 1          # kernels after the previous step's adamw .. this step's adamw
        step = rows[lo:hi]
        phase = "augment"
        cur = {p: [] for p in PHASES}
        seen_loss = False
        for s, e, nm in step:
            if phase == "augment" and "conv0_fwd" in nm:
                phase = "forward"
            elif phase == "forward" and ("supcon" in nm or "nll" in nm):
                phase = "losses"
            elif phase == "losses" and ("supcon_gsym" in nm or "supcon_bwd" in nm or "utt_head_bwd" in nm or "meanpool_bwd" in nm):
                phase = "backward"          # total.backward() starts with the losses' own backward kernels
            elif "adamw_kernel" in nm:
                phase = "optimizer"
            cur[phase].append((s, e, nm))
        for p, ks in cur.items():
            if not ks:
                continue
            agg[p]["n"] += len(ks)
            agg[p]["k"] += sum(e - s for s, e, _ in ks) / 1e6
            agg[p]["span"] += (max(e for _, e, _ in ks) - min(s for s, _, _ in ks)) / 1e6
        prev_e, prev_nm = None, None
        for s, e, nm in step:
            if prev_e is not None and s > prev_e:
                g = gaps_after[prev_nm]; g[0] += 1; g[1] += (s - prev_e) / 1e3
            if prev_e is None or e > prev_e:
                prev_e, prev_nm = e, nm
    evs = json.load(open(events_json)) if events_json else None
    print("# default train step (configs[2]), %d steady-state steps of the rocprofv3 kernel trace; un-profiled spans: HIP events at the phase boundaries" % n)
    print("%-10s %9s %12s %14s %16s %22s" % ("phase", "launches", "kernel ms", "span ms (prof)", "span ms (events)", "events span - kernel ms"))
    tk = ts = te = 0.0
    for p, a in agg.items():
        k, sp = a["k"] / n, a["span"] / n
        es = evs["spans_ms"][p] if evs and "spans_ms" in evs else None
        # the events' "losses" span holds only the losses' forward; their backward kernels run inside the backward span
        print("%-10s %9.1f %12.3f %14.3f %16s %22s" % (p, a["n"] / n, k, sp, "%.3f" % es if es is not None else "-", "%.3f" % (es - k) if es is not None else "-"))
        tk += k; ts += sp; te += es or 0.0
    print("%-10s %9.1f %12.3f %14.3f %16s %22s" % ("sum", sum(a["n"] for a in agg.values()) / n, tk, ts, "%.3f" % te if evs else "-", "%.3f" % (te - tk) if evs else "-"))
    if evs:
        print("# un-profiled: %.3f ms per step on the host clock (all %d steps after a synchronize), %.3f ms from the first to the last event of a step (steps 2..), %.3f ms between the last event of a step and the first of the next"
              % (evs["ms_per_step_wall"], evs.get("steps", 0), evs.get("step_span_ms", 0.0), evs.get("between_steps_ms", 0.0)))
        if "first_step_spans_ms" in evs:
            print("# the FIRST step after the synchronize (GPU idle while the launch thread prepares the step; not in the table): "
                  + ", ".join("%s %.3f" % (p, evs["first_step_spans_ms"][p]) for p in PHASES))
    print("# idle time under the profiler by the kernel that PRECEDES the gap (per step):")
    for nm, (c, us) in sorted(gaps_after.items(), key=lambda kv: -kv[1][1])[:14]:
        print("  %-62s %6.1f gaps %8.1f us" % (nm[:62], c / n, us / n))


if __name__ == "__main__":
    if sys.argv[1] == "events":
        run_events()
    else:
        run_trace(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
