"""ISA audit for kernels that read MFMA fragments with inline-asm `ds_read_b64_tr_b16` (frag_t_raw, csrc/gemm_common.h; pw_read,
csrc/posconv.hip): those reads are asynchronous and invisible to the compiler's lgkmcnt bookkeeping, so nothing may touch their
destination registers before an explicit `s_waitcnt lgkmcnt(0)`.  The script compiles the given .hip files to gfx950 assembly and walks
every kernel (loops twice): a vector instruction that names the destination of an LDS read still in flight is reported.
(Round 3: it found an MFMA of scl_gemm_dma_kernel / scl_gemm_big_kernel that the scheduler had lifted above a bare
`asm volatile("s_waitcnt lgkmcnt(0)")`; the wait now takes the fragments as read-write operands.)  Kernels that use the BUILTIN tr read
(attention.hip) are tracked by the compiler, wait with counted lgkmcnt(N), and are reported as false positives here.

    python tools/isa_hazard_audit.py [csrc/gemm.hip csrc/gemm_w8.hip csrc/gemm_x2.hip csrc/posconv.hip]
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "scl-deepfake-audio-detection_amd", "csrc")


def regs(tok):
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


def audit(asm_text):
    """LDS operations retire in order, so `s_waitcnt lgkmcnt(N)` guarantees everything but the N youngest operations (scalar loads share
    the counter and return out of order: every one still in flight is assumed to be among the completed ones — the conservative side).
    Every kernel is walked in program order, and every backward branch is followed once more with the state at the branch, so that
    what a loop iteration leaves in flight meets the waits at the top of the next one."""
    bad = 0
    for km in re.finditer(r"^(_Z\S+):[^\n]*\n(.*?)\.Lfunc_end", asm_text, re.S | re.M):
        name, body = km.group(1), km.group(2)
        if "ds_read_b64_tr_b16" not in body:
            continue
        prog, labels = [], {}
        for line in body.splitlines():
            line = line.split(";")[0].strip()
            if not line or line.startswith("."):
                if line.endswith(":"):
                    labels[line[:-1]] = len(prog)
                continue
            if line.endswith(":"):
                labels[line[:-1]] = len(prog)
                continue
            prog.append(line)
        queue, hazards, n_tr, examples, taken = [], 0, 0, [], set()      # queue: (kind, registers) oldest first
        pc = 0
        while pc < len(prog):
            line = prog[pc]
            parts = line.replace(",", " ").split()
            op, args = parts[0], parts[1:]
            pc += 1
            if op.startswith("ds_"):
                dst = regs(args[0]) if op.startswith("ds_read") else set()
                if op == "ds_read_b64_tr_b16":
                    n_tr += 1
                queue.append(("lds", dst))
                continue
            if op.startswith("s_load") or op.startswith("s_buffer_load") or op in ("s_memtime", "s_memrealtime"):
                queue.append(("smem", set()))
                continue
            if op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", line)
                if m:
                    n = int(m.group(1))
                    if n == 0:
                        queue = []
                    else:
                        done = len(queue) - n - sum(1 for k, _ in queue if k == "smem")
                        while done > 0 and queue:
                            k = next((x for x, e in enumerate(queue) if e[0] == "lds"), None)
                            if k is None:
                                break
                            queue.pop(k)
                            done -= 1
                continue
            if op.startswith("s_cbranch") or op == "s_branch":
                tgt = args[-1]
                if tgt in labels and labels[tgt] < pc and (pc, tgt) not in taken:
                    taken.add((pc, tgt))
                    pc = labels[tgt]
                continue
            if op.startswith("s_"):
                continue
            used = set()
            for t in args:
                used |= regs(t)
            pending = set()
            for k, r in queue:
                pending |= r
            if used & pending:
                hazards += 1
                if len(examples) < 2:
                    examples.append(line)
        print("%-72s tr reads %4d  uses of a pending register: %d %s" % (name[-72:], n_tr, hazards, examples if hazards else ""))
        bad += hazards
    return bad


def main():
    files = sys.argv[1:] or [os.path.join(CSRC, f) for f in ("gemm.hip", "gemm_w8.hip", "gemm_x2.hip", "posconv.hip")]
    total = 0
    with tempfile.TemporaryDirectory() as td:
        for f in files:
            out = os.path.join(td, os.path.basename(f) + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "--cuda-device-only", "-S", "-o", out, f],
                           check=True, stderr=subprocess.DEVNULL)
            print("==", os.path.basename(f))
            total += audit(open(out).read())
    print("total:", total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
