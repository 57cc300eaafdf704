"""One row per GEMM class (M, N, K, epilogue) of the bench step: calls per step, live launch duration (the same HIP-event pairs bench.py's
roofline sums), TFLOP/s, and — joined by dispatch order from two rocprofv3 --pmc passes of the same command — FETCH_SIZE (x 2, gfx950
note of MI355X_MICROARCH.md) and WRITE_SIZE per launch next to the algorithmic operand / output bytes.

    python tools/gemm_classes.py LAUNCHES.json [--fetch DIR_OR_CSV] [--write DIR_OR_CSV] [--log-fetch L.json] [--log-write L.json] > profiles/r6_gemm_classes.txt

LAUNCHES.json comes from `python bench.py --dump-gemm-launches LAUNCHES.json`; each --pmc pass writes its OWN launch log (`--log-*`), whose
order is joined with the GEMM-family dispatches of that pass's counter_collection.csv (the last len(log) of them).
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import sys

F = {"A_T": 1, "B_T": 2}      # filled from include/scl_hip.h below


def header_flags():
    hdr = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "scl_hip.h")).read()
    out = {}
    for name, val in re.findall(r"#define\s+SCL_GEMM_([A-Z0-9_]+)\s+\(?\s*(1\s*<<\s*\d+|0x[0-9a-fA-F]+|\d+)\s*\)?", hdr):
        out[name] = eval(val)
    return out


def classify(meta, H):
    M, N, K, flags, z, variant = meta[:6]
    if variant == 4:
        return ("grouped wgrad (A^T B, whole-K tiles)", "%d tiles, %d members" % (flags, z), None)
    if variant == 5:
        kind = {1: "pos-conv fwd", 0: "pos-conv dgrad", 2: "pos-conv wgrad"}[flags]
        return (kind, "16 groups", None)
    at, bt = bool(flags & H["A_T"]), bool(flags & H["B_T"])
    act = (flags >> H["ACT_SHIFT"]) & 0xF
    rmode = (flags >> H["RMODE_SHIFT"]) & 0xF
    epi = []
    if flags & H["HAS_BIAS"]:
        epi.append("bias")
    if act:
        epi.append({5: "gelu+gelu'"}.get(act, "act%d" % act))
    if flags & H["HAS_C2"]:
        epi.append("2nd out")
    if rmode == 1:
        epi.append("+R" + ("f32" if flags & H["R_F32"] else "bf16"))
    if rmode == 2:
        epi.append("xR")
    if flags & H["DROPOUT"]:
        epi.append("drop")
    epi.append("f32 out" if flags & H["C_F32"] else "bf16 out")
    lay = ("T" if at else "N") + ("T" if bt else "N")
    return ("%s z=%d" % (lay, z), ", ".join(epi), (at, bt))


def alg_bytes(meta, H):
    M, N, K, flags, z, variant = meta[:6]
    if variant == 4:
        # grouped weight gradients: `flags` tiles of 256 x 256 out of the 192 a transformer layer has (fc2 64, fc1 64, out 16, qkv 48); a
        # layer's eight operand matrices are 16384 columns of bf16 over K rows, its four outputs 12.6 M f32
        tiles = flags
        return tiles / 192.0 * 16384 * K * 2, tiles * 256 * 256 * 4
    if variant == 5:
        # pos-conv (16 groups x 64 channels, 128 taps): M = B T rows (fwd / dgrad) or K = B T (wgrad); padded slab rows ~ M (1 + 128 / 199)
        if flags == 2:
            return 2 * K * N * 2 * 1.64, M * N * 4 / 1.0
        return M * N * 2 * 1.64 + K * N // 16 * 2 + M * N * 4, M * N * 4 + (M * N * 2 if flags == 1 else 0)
    if variant == 6:
        variant = 2
    zz = max(1, z)
    out = M * N * (4 if flags & H["C_F32"] else 2) * zz
    rd = (M * K + N * K) * 2 * (zz if zz > 1 and not (flags & H["C_F32"]) else 1)
    if flags & H["HAS_C2"]:
        out += M * N * (4 if flags & H["C2_F32"] else 2)
    if (flags >> H["RMODE_SHIFT"]) & 0xF:
        rd += M * N * (4 if flags & H["R_F32"] else 2)
    return rd, out


GEMM_RE = re.compile(r"scl_gemm_(w8s_group_|w8s_|w8_|dma_|w8p_|x2_)?kernel|posconv_mfma_kernel|posconv_wgrad_kernel")


def pmc_per_launch(path, counter, log):
    """mean counter value per launch, keyed by launch index of `log` (dispatch order join)"""
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return None
    rows = []
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != counter or not GEMM_RE.search(r["Kernel_Name"]) or "f32" in r["Kernel_Name"].split("(")[0]:
            continue
        rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), r["Kernel_Name"]))
    rows.sort()
    # a dispatch may appear once per XCD / dimension: sum per dispatch id
    per = collections.OrderedDict()
    for d, v, k in rows:
        per.setdefault(d, [0.0, k])[0] += v
    vals = list(per.values())
    n = len(log)
    if len(vals) < n:
        print("# %s: %d GEMM dispatches in the counter file < %d logged launches: no join" % (counter, len(vals), n), file=sys.stderr)
        return None
    return [v for v, _ in vals[-n:]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("launches")
    ap.add_argument("--fetch"), ap.add_argument("--write"), ap.add_argument("--log-fetch"), ap.add_argument("--log-write")
    a = ap.parse_args()
    H = header_flags()
    d = json.load(open(a.launches))
    nstep = d["profiled_steps"]
    L = d["launches"]
    cls = collections.OrderedDict()

    def key(meta):
        c = classify(meta, H)
        return (tuple(meta[:3]), c[0], c[1])

    for row in L:
        ms, meta = row[0], row[1:]
        e = cls.setdefault(key(meta), {"n": 0, "ms": 0.0, "meta": meta, "fetch": [], "write": []})
        e["n"] += 1
        e["ms"] += ms
    for name, path, logp in (("fetch", a.fetch, a.log_fetch), ("write", a.write, a.log_write)):
        if not path or not logp:
            continue
        lg = json.load(open(logp))["launches"]
        vals = pmc_per_launch(path, "FETCH_SIZE" if name == "fetch" else "WRITE_SIZE", lg)
        if vals is None:
            continue
        for row, v in zip(lg, vals):
            k = key(row[1:])
            if k in cls:
                cls[k][name].append(v * 1024.0 * (2.0 if name == "fetch" else 1.0))      # KiB -> bytes; FETCH_SIZE doubled (gfx950)
    tot_ms = sum(e["ms"] for e in cls.values())
    tot_fl = 0.0
    fam_meas = fam_alg = 0.0
    print("# GEMM family of `python bench.py` (batch %d x %d, %s), %d profiled steps, sources %s" % (d["batch"], d["samples"], d["model"], nstep, d["gemm_src_sha"]))
    print("# us = mean launch duration from the launch's own dispatch time stamps (what roofline.frac sums); FETCH = FETCH_SIZE x 2 (gfx950), "
          "WRITE = WRITE_SIZE, from separate --pmc passes joined by dispatch order; alg = unique operand bytes read / output bytes written")
    print("%-22s %-40s %7s %6s %6s | %5s %8s %8s %7s | %8s %8s %8s %8s %6s" % ("layout", "epilogue", "M", "N", "K", "calls", "us", "ms/step", "TFLOP/s",
                                                                              "FETCH MB", "alg rd", "WRITE MB", "alg wr", "ratio"))
    for k, e in sorted(cls.items(), key=lambda kv: -kv[1]["ms"]):
        (M, N, K), lay, epi = k
        meta = e["meta"]
        if meta[5] == 4:
            fl = 2.0 * 256 * 256 * K * meta[3]      # `flags` = tiles of the launch
        else:
            fl = 2.0 * M * N * K * (max(1, meta[4]) if meta[5] < 4 and not (meta[3] & H["C_F32"] and meta[4] > 1) else 1)
            if meta[5] < 4 and (meta[3] & H["C_F32"]) and meta[4] > 1:
                fl = 2.0 * M * N * K      # split-K: z slabs share one contraction (batched launches with f32 output are counted once: under-estimate, flagged)
        us = e["ms"] / e["n"] * 1e3
        ab = alg_bytes(meta, H)
        fe = sum(e["fetch"]) / len(e["fetch"]) / 1e6 if e["fetch"] else None
        wr = sum(e["write"]) / len(e["write"]) / 1e6 if e["write"] else None
        ratio = ((fe or 0) + (wr or 0)) / ((ab[0] + ab[1]) / 1e6) if (ab and fe is not None and wr is not None) else None
        fmt = lambda v, f="%8.1f": (f % v) if v is not None else " " * (int(re.search(r"%(\d+)", f).group(1)) - 1) + "-"
        print("%-22s %-40s %7d %6d %6d | %5.1f %8.1f %8.3f %7s | %s %s %s %s %s" % (
            lay, epi, M, N, K, e["n"] / nstep, us, e["ms"] / nstep, ("%7.0f" % (fl / us / 1e6)) if fl else "      -",
            fmt(fe), fmt(ab[0] / 1e6 if ab else None), fmt(wr), fmt(ab[1] / 1e6 if ab else None), fmt(ratio, "%6.2f")))
        if fl:
            tot_fl += fl * e["n"]
        if ab and fe is not None and wr is not None:
            fam_meas += (fe + wr) * e["n"] / nstep
            fam_alg += (ab[0] + ab[1]) / 1e6 * e["n"] / nstep
    print("# total %.3f ms of GEMM-family launches per step, %d launches per step" % (tot_ms / nstep, len(L) // nstep))
    if fam_alg > 0:
        nl = len(L) // nstep
        print("# family traffic per step: measured (FETCH x 2 + WRITE) %.1f GB vs algorithmic %.1f GB = %.2f x; per launch %.0f MB vs %.0f MB"
              % (fam_meas / 1e3, fam_alg / 1e3, fam_meas / fam_alg, fam_meas / nl, fam_alg / nl))


if __name__ == "__main__":
    main()
