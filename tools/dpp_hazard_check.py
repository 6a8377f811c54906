"""The inline-assembly DPP stages of k_osd (csrc/kernels/osd.hpp: OSD_DPP_STAGE) read their partner lane through a DPP operand.  gfx9 needs
two wait states between a VALU write of a VGPR and a DPP read of it; LLVM's hazard recognizer inserts them for code it schedules, but it
does not look INSIDE an inline-asm block, so an asm block that starts right behind the producer of its first DPP source would silently
sort wrongly after a compiler update.  This scans the disassembly of the shipped library:

    python tools/dpp_hazard_check.py [lib.so]      -> exit code 1 and a listing if any DPP read follows a VALU write of its source
                                                     register with fewer than two wait states in between

(every instruction between the two counts one wait state, `s_nop N` counts N + 1).  tests/test_host_layer.py runs it on both builds."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
DPP = re.compile(r"\b(quad_perm|row_shl|row_shr|row_ror|wave_shl|wave_shr|wave_rol|wave_ror|row_mirror|row_half_mirror|row_bcast|row_newbcast|row_share|row_xmask)\b")


def regs(tok):
    """v12 -> {12}; v[4:5] -> {4, 5}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def scan(lib, only=("k_osd",)):
    """-> (number of DPP instructions seen, [violations])"""
    n_dpp, bad = 0, []
    with tempfile.TemporaryDirectory() as d:
        tmp = os.path.join(d, os.path.basename(lib))
        os.symlink(os.path.abspath(lib), tmp)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", tmp], check=True, capture_output=True, cwd=d)
        for f in sorted(x for x in os.listdir(d) if "amdgcn" in x):
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(d, f)], check=True,
                                 capture_output=True, text=True).stdout
            fn, hist = None, []                  # hist: the last instructions of the current function as (mnemonic, dest regs, wait states)
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
                if m:
                    fn, hist = m.group(1), []
                    continue
                if fn is None or not any(o in fn for o in only):
                    continue
                ins = line.split("//")[0].strip()
                if not ins or ins.endswith(":"):
                    continue
                parts = ins.replace(",", " ").split()
                mn, ops = parts[0], parts[1:]
                ws = 1
                if mn == "s_nop":
                    ws = int(ops[0], 0) + 1 if ops else 1
                dest = regs(ops[0]) if (mn.startswith("v_") and ops and not mn.startswith("v_cmp") and not mn.startswith("v_readlane") and not mn.startswith("v_readfirstlane")) else set()
                if DPP.search(ins) and mn.startswith("v_"):
                    n_dpp += 1
                    # the DPP operand is src0: the first source (operand 1, or operand 2 when a VOPC-style sdst / vcc comes first)
                    srcs = [o for o in ops[1:] if regs(o)]
                    src0 = regs(srcs[0]) if srcs else set()
                    waited = 0
                    for pm, pd, pw in reversed(hist):
                        if waited >= 2:
                            break
                        if pd & src0:
                            bad.append(f"{fn}: `{ins}` reads {sorted(src0)} {waited} wait state(s) after `{pm}`")
                            break
                        waited += pw
                hist.append((ins, dest, ws))
                hist = hist[-6:]
    return n_dpp, bad


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pyft8_amd", "libft8rx.so")
    n, bad = scan(lib)
    print(f"{os.path.basename(lib)}: {n} DPP instructions in k_osd*, {len(bad)} read a register written fewer than 2 wait states before")
    for b in bad:
        print("  " + b)
    sys.exit(1 if bad or n == 0 else 0)
