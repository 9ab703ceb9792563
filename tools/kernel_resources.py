#!/usr/bin/env python3
"""Per-kernel resource table of one of the HIP sources, compiled for gfx950 with the repo's flags:
VGPRs, SGPRs, spills, scratch, LDS, workgroup size, MFMA / LDS-DMA / ds_read / `s_waitcnt vmcnt(0)` counts.

    python tools/kernel_resources.py smart-nar_fast_tts_amd/csrc/gemm_conv.hip [--grep conv_gemm] [--rev HEAD]

Compiles device-only to assembly (hipcc cross-compiles without a GPU), reads the kernels' `amdhsa.kernels` metadata and
counts instructions per kernel body.  Used after every edit of a hot kernel: an instantiation that starts to spill, or
whose VGPR count crosses an occupancy step, shows up here before it shows up as time on the GPU box.  `--rev R` compiles the
file as of git revision R instead (for a before / after diff)."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-result", "--cuda-device-only", "-S"]
EXTRA = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("--grep", default="")
    ap.add_argument("--rev", default=None)
    ap.add_argument("--assert-no-spill", action="store_true", help="exit 1 when any listed kernel spills a register or uses scratch (hot-path hygiene gate)")
    ap.add_argument("-I", dest="inc", action="append", default=[], help="extra include directory (lab harnesses: -I smart-nar_fast_tts_amd/csrc)")
    a = ap.parse_args()
    src = os.path.abspath(a.src)
    with tempfile.TemporaryDirectory() as tmp:
        inc = os.path.dirname(src)
        if a.rev:
            rel = os.path.relpath(src, ROOT)
            d = os.path.join(tmp, "rev")
            os.makedirs(d)
            for f in subprocess.run(["git", "ls-tree", "--name-only", a.rev, os.path.dirname(rel) + "/"], capture_output=True, text=True, cwd=ROOT).stdout.split():
                if f.endswith((".h", ".hip")):
                    open(os.path.join(d, os.path.basename(f)), "w").write(subprocess.run(["git", "show", f"{a.rev}:{f}"], capture_output=True, text=True, cwd=ROOT).stdout)
            os.makedirs(os.path.join(tmp, "include"))
            # (api.hip includes ../../include/nar_fs2.h; the kernel sources do not)
            src, inc = os.path.join(d, os.path.basename(src)), d
        out = os.path.join(tmp, "k.s")
        cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + EXTRA.get(os.path.basename(src), []) + ["-I", inc] + [x for i in a.inc for x in ("-I", os.path.abspath(i))] + [src, "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stdout + r.stderr)
        asm = open(out).read()
    # instruction counts per kernel body
    counts, name = {}, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1)
            counts[name] = {"mfma": 0, "dma": 0, "ds_read": 0, "waitvm0": 0}
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            name = None if line.startswith(".Lfunc_end") else name
        if name is None or not line.startswith("\t"):
            continue
        t = line.strip()
        if t.startswith("v_mfma"):
            counts[name]["mfma"] += 1
        elif t.startswith("buffer_load") and " lds" in t:
            counts[name]["dma"] += 1
        elif t.startswith("ds_read") or t.startswith("ds_load"):
            counts[name]["ds_read"] += 1
        elif t.startswith("s_waitcnt vmcnt(0)"):
            counts[name]["waitvm0"] += 1
    # metadata (YAML at the end of the file)
    kern, cur = [], None
    for line in asm.splitlines():
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip().strip("'")
        if k == "agpr_count":
            cur = {"agpr": v}
            kern.append(cur)
        elif cur is not None and k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                                      "group_segment_fixed_size", "max_flat_workgroup_size"):
            cur[k] = v
        elif cur is not None and k == "name" and "name" not in cur and not v.endswith(".kd") and v.startswith("_Z"):
            cur["name"] = v
    rows = [k for k in kern if "name" in k]
    dm = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in rows), capture_output=True, text=True).stdout.splitlines()
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'vspill':>6} {'sspill':>6} {'scratch':>7} {'lds':>7} {'wg':>5} {'mfma':>5} {'dma':>4} {'dsrd':>5} {'vm0':>4}  kernel")
    bad = []
    for k, d in sorted(zip(rows, dm), key=lambda x: x[1]):
        if a.grep and a.grep not in d:
            continue
        if any(int(k.get(f, 0) or 0) for f in ("vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size")):
            bad.append(d)
        c = counts.get(k["name"], {})
        short = re.sub(r"\(.*", "", d).replace("void ns::", "")
        print(f"{k.get('vgpr_count', '?'):>5} {k.get('agpr', '?'):>5} {k.get('sgpr_count', '?'):>5} {k.get('vgpr_spill_count', '?'):>6} {k.get('sgpr_spill_count', '?'):>6} "
              f"{k.get('private_segment_fixed_size', '?'):>7} {k.get('group_segment_fixed_size', '?'):>7} {k.get('max_flat_workgroup_size', '?'):>5} "
              f"{c.get('mfma', '?'):>5} {c.get('dma', '?'):>4} {c.get('ds_read', '?'):>5} {c.get('waitvm0', '?'):>4}  {short}")
    if a.assert_no_spill and bad:
        sys.exit("kernels that spill or use scratch:\n  " + "\n  ".join(bad))


if __name__ == "__main__":
    main()
