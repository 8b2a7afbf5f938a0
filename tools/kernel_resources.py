#!/usr/bin/env python3
"""Per-kernel resources of a built library, read from the code objects inside it (nothing is loaded, no GPU): registers, LDS, scratch,
and the wavefronts per SIMD they allow on gfx950 (512 VGPRs per SIMD lane, at most 8 wavefronts).

    python tools/kernel_resources.py [pl-slam_amd/libplslam_hip.so]

The library's .hip_fatbin section holds one clang offload bundle per translation unit; each is unbundled with clang-offload-bundler and
its AMDGPU metadata note parsed (llvm-readelf --notes).  tests/test_kernel_resources.py asserts the occupancy points DESIGN.md relies on."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
FIELDS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".group_segment_fixed_size", ".private_segment_fixed_size", ".uses_dynamic_stack",
          ".vgpr_spill_count", ".sgpr_spill_count", ".max_flat_workgroup_size")


def unified_vgprs(rec):
    """Registers of the unified file a wavefront takes: architectural VGPRs (rounded up to 4 when accumulation registers follow them)
    plus AGPRs -- the compiler parks spilled values in AGPRs before it goes to scratch, and they count against the occupancy."""
    v, a = int(rec.get("vgpr_count", 0)), int(rec.get("agpr_count", 0))
    return v if a == 0 else (v + 3) // 4 * 4 + a


def waves_per_simd(vgprs):
    """Wavefronts of `vgprs` unified registers a gfx950 SIMD holds (allocation granule 8, 512 per lane, at most 8 wavefronts)."""
    g = max(8, (int(vgprs) + 7) // 8 * 8)
    return min(8, 512 // g)


def kernel_resources(lib):
    """{demangled short kernel name: {field: value}} for every kernel of every gfx950 code object in `lib`."""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(td, "copy.so")])
        d = open(fat, "rb").read()
        pos = [m.start() for m in re.finditer(re.escape(MAGIC), d)]
        for i, p in enumerate(pos):
            e = pos[i + 1] if i + 1 < len(pos) else len(d)
            b, co = os.path.join(td, "b%d.bin" % i), os.path.join(td, "co%d.o" % i)
            open(b, "wb").write(d[p:e])
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + b, "--targets=" + TARGET, "--output=" + co],
                               capture_output=True)
            if r.returncode != 0 or not os.path.exists(co):
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
            # one YAML map per kernel under amdhsa.kernels: split at the list items
            for blk in re.split(r"\n\s+- \.", notes):
                m = re.search(r"(?:^|\n)\s*\.?name:\s+(\S+)", blk)
                if not m or ".vgpr_count" not in ("." + blk):
                    continue
                name = m.group(1)
                sym = re.search(r"\.symbol:\s+(\S+)", blk)
                if sym:
                    name = sym.group(1).replace(".kd", "")
                rec = {}
                for f in FIELDS:
                    mm = re.search(re.escape(f[1:]) + r":\s+(\S+)", blk)
                    if mm:
                        v = mm.group(1)
                        rec[f[1:]] = (v == "true") if v in ("true", "false") else int(v)
                short = re.sub(r"^_ZN3plh\d+", "", name)
                short = re.match(r"[A-Za-z_0-9]+?(?=E[A-Z]|E$|I[A-Z]|ILi)", short).group(0) if re.match(r"[A-Za-z_0-9]+?(?=E[A-Z]|E$|I[A-Z]|ILi)", short) else short
                key = short
                n = 2
                while key in out:   # template instances share the short name
                    key = "%s#%d" % (short, n)
                    n += 1
                rec["symbol"] = name
                out[key] = rec
    return out


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pl-slam_amd", "libplslam_hip.so")
    res = kernel_resources(lib)
    print("%-28s %5s %5s %5s %7s %8s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "lds", "scratch", "waves"))
    for k in sorted(res, key=lambda k: -unified_vgprs(res[k])):
        r = res[k]
        print("%-28s %5d %5d %5d %7d %8d %6d" % (k[:28], r.get("vgpr_count", 0), r.get("agpr_count", 0), r.get("sgpr_count", 0),
                                                r.get("group_segment_fixed_size", 0), r.get("private_segment_fixed_size", 0),
                                                waves_per_simd(unified_vgprs(r))))


if __name__ == "__main__":
    main()
