#!/usr/bin/env python3
"""tools/isa_census.py [kernel-name-substring ...] -- static instruction census of the gfx950 kernels of libhvk:
hipcc -S of hvk_direct.hip / hvk_kernels.hip / hvk_secam.hip (device only), then per kernel the number of
instructions by class (VALU half rate / full rate, matrix unit, SALU, LDS, vector memory, waits, branches) and the
most frequent opcodes. Static counts: loops and skipped branches are not weighed -- the dynamic totals are the
SQ_INSTS_* counters in profiles/*_pmc_counters.json. Runs in the build container (no GPU needed).

  python tools/isa_census.py > profiles/r03_isa_census.txt
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "hacktv_amd", "csrc")
DEFAULT = ["hvk_k_direct<1, 1, 1>", "hvk_k_direct<0, 1, 1>", "hvk_k_prep<13, 1024, 0>", "hvk_k_prep<13, 1024, 1>",
           "hvk_k_raster<13, 0, 0, 0, 1024, 0>", "hvk_k_filter<51, 3, 0, 1, 1>", "hvk_k_secam_chain", "hvk_k_secam_cells"]

# issue cost per wave64 instruction measured with tools/ubench_valu.hip (profiles/r03_valu_issue_rates.txt): 4 cycles
HALF = ("v_dot2", "v_dot4", "v_mad_", "v_mul_lo", "v_mul_hi", "v_perm", "v_alignbit", "v_alignbyte", "v_pk_", "v_bfe", "v_bfi", "v_mad_u64", "v_mad_i64",
        "v_cvt_pk", "v_lshl_add_u64", "v_add3", "v_lshl_or", "v_and_or", "v_or3", "v_xad", "v_bitop3", "v_lshl_add_u32", "v_add_lshl", "v_rcp", "v_sad", "v_min3", "v_max3", "v_med3")


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        if "f64" in op:
            return "valu f64"
        return "valu 4-cycle" if op.startswith(HALF) else "valu 2-cycle"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
        return "wait/nop/barrier"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    want = sys.argv[1:] or DEFAULT
    with tempfile.TemporaryDirectory() as tmp:
        for src in ("hvk_direct.hip", "hvk_fused.hip", "hvk_kernels.hip", "hvk_secam.hip"):
            out = os.path.join(tmp, src + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I../../include", "-I.",
                            "--cuda-device-only", "-S", src, "-o", out], cwd=SRC, check=True, stderr=subprocess.DEVNULL)
            text = open(out).read()
            for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)\n\.Lfunc_end", text, re.S | re.M):
                name = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
                short = re.sub(r"^void ", "", name.split("(")[0])
                if not any(w in short for w in want):
                    continue
                ops = [ln.split()[0] for ln in m.group(2).splitlines() if ln.startswith("\t") and ln.strip() and not ln.strip().startswith((";", "."))]
                cls = collections.Counter(classify(o) for o in ops)
                top = collections.Counter(o for o in ops if o.startswith(("v_", "ds_", "global_")))
                meta = re.search(r"\.amdhsa_kernel " + re.escape(m.group(1)) + r"\n(.*?)\.end_amdhsa_kernel", text, re.S)
                vg = re.search(r"next_free_vgpr (\d+)", meta.group(1)).group(1) if meta else "?"
                lds = re.search(r"group_segment_fixed_size (\d+)", meta.group(1)).group(1) if meta else "?"
                scr = re.search(r"private_segment_fixed_size (\d+)", meta.group(1)).group(1) if meta else "?"
                print("%s\n  %d instructions, %s VGPRs, %s B LDS (static), %s B scratch" % (short, len(ops), vg, lds, scr))
                print("  " + ", ".join("%s %d" % kv for kv in sorted(cls.items(), key=lambda kv: -kv[1])))
                print("  most frequent: " + ", ".join("%s %d" % kv for kv in top.most_common(14)))
                print()


if __name__ == "__main__":
    main()
