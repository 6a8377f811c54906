"""Per-kernel resource table of libft8rx.so from the code-object notes: VGPRs, SGPRs, LDS, scratch, and the waves per SIMD they allow.

    python tools/kernel_resources.py [lib.so] > profiles/rNN_kernel_resources.txt

The library holds one device code object per translation unit (ft8rx.hip, ft8rx_ilp.hip); every kernel must appear exactly once
(tests/test_host_layer.py::test_every_kernel_exists_once).  No GPU needed: llvm-objdump unbundles, llvm-readelf reads the notes."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(lib):
    """-> [(code object index, demangled-ish name, vgprs, sgprs, lds bytes, scratch bytes)]"""
    out = []
    with tempfile.TemporaryDirectory() as d:
        tmp = os.path.join(d, os.path.basename(lib))
        os.symlink(os.path.abspath(lib), tmp)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", tmp], check=True, capture_output=True, cwd=d)
        objs = sorted(f for f in os.listdir(d) if "amdgcn" in f)
        for i, f in enumerate(objs):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(d, f)], check=True, capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count")[1:] if "- .agpr_count" in notes else re.split(r"\n\s+- \.", notes)[1:]:
                get = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk)
                if not get("name") or not get("vgpr_count"):
                    continue
                name = get("name").group(1)
                try:
                    name = re.sub(r"^void ", "", subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()).split("(")[0]
                except OSError:
                    pass
                out.append((i, name, int(get("vgpr_count").group(1)), int(get("sgpr_count").group(1)),
                            int(get("group_segment_fixed_size").group(1)), int(get("private_segment_fixed_size").group(1))))
    return out


def waves_per_simd(vgprs, lds, threads=None):
    """gfx950: 512 VGPRs per SIMD lane (allocation granule 8), 160 KB LDS per CU; at most 8 waves per SIMD."""
    by_reg = min(8, 512 // max(8, (vgprs + 7) // 8 * 8))
    return by_reg


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pyft8_amd", "libft8rx.so")
    ks = kernels(lib)
    print(f"# {os.path.basename(lib)}: {len(ks)} kernels in {len(set(k[0] for k in ks))} code objects (0 = ft8rx.hip, 1 = ft8rx_ilp.hip)")
    print(f"{'unit':>4} {'kernel':<44} {'VGPRs':>6} {'SGPRs':>6} {'LDS B':>7} {'scratch B':>9} {'waves/SIMD by VGPRs':>20}")
    for u, n, v, s, l, p in sorted(ks, key=lambda k: (k[0], k[1])):
        print(f"{u:>4} {n:<44} {v:>6} {s:>6} {l:>7} {p:>9} {waves_per_simd(v, l):>20}")
    names = [k[1] for k in ks]
    dup = sorted({n for n in names if names.count(n) > 1})
    print("# duplicates:", ", ".join(dup) if dup else "none")
