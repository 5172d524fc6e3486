"""Post-compile checks on the gfx950 machine code of the kernel library (build-time tooling, no GPU needed).

``store_data_hazards`` looks for one instruction pattern hipcc (ROCm 7.2) emits and the hardware does not tolerate:

    buffer_store_dwordx4 v[a:a+3], vOFF, s[..], sN offen     ; soffset in an SGPR
    v_xxx            v[a..a+3], ...                          ; vector write of a data register in the next slot(s)

gfx950 reads the data registers of a > 64-bit buffer store a few cycles after issue.  LLVM's hazard recogniser inserts the
required wait state only when soffset is a constant (the ISA manual exempts the SGPR form); on the SGPR form the vector
write wins the race for the lanes read last (lanes 12-15 of every 16, first data register) -- seen as 16 stray dwords per
wave in about one FIRST call in ten of a fresh process (cold instruction cache) of the phased 16-bit convolution, and it is
how a float16 training run picked up its first non-finite activation.  The kernels fence such
stores (scheduling barriers + ``s_nop 2``); this check keeps it that way for every object that is built.
"""
import os
import re
import shutil
import subprocess
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"

_STORE = re.compile(r"^\s*buffer_store_(?:dwordx[34]|format_xyzw?|format_d16_xyzw)\s+v\[(\d+):(\d+)\],\s*[^,]+,\s*s\[\d+:\d+\],\s*(\S+)")
_VDST = re.compile(r"^\s*(v_\w+)\s+(?:v(\d+)|v\[(\d+):(\d+)\])")
_NOP = re.compile(r"^\s*s_nop\s+(\d+)")
_NO_VGPR_DST = ("v_cmp", "v_readlane", "v_readfirstlane", "v_nop")
MIN_WAIT_STATES = 2          # instructions (or s_nop cycles) between the store and a vector write of its data


def _tool(name):
    exe = os.path.join(LLVM_BIN, name)
    return exe if os.path.exists(exe) else shutil.which(name)


def tools_available():
    return all(_tool(t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump"))


def disassemble(obj_path):
    """gfx950 disassembly (text) of the device code embedded in a hipcc object file."""
    with tempfile.TemporaryDirectory() as tmp:
        fat, elf = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "co.elf")
        subprocess.run([_tool("llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj_path, fat], check=True)
        subprocess.run([_tool("clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat, "--targets=" + TARGET,
                        "--output=" + elf], check=True, capture_output=True)
        return subprocess.run([_tool("llvm-objdump"), "-d", "--mcpu=gfx950", elf], check=True, capture_output=True, text=True).stdout


def store_data_hazards(text, min_wait_states=MIN_WAIT_STATES):
    """[(symbol, store line, offending line)] for every wide buffer store with an SGPR soffset whose data registers are
    written by a vector instruction fewer than `min_wait_states` slots later."""
    lines = text.splitlines()
    found, symbol = [], ""
    for i, raw in enumerate(lines):
        line = raw.split("//")[0].rstrip()
        if line.endswith(">:") and "<" in line:
            symbol = line[line.index("<") + 1:-2]
            continue
        m = _STORE.match(line)
        if not m or not m.group(3).startswith(("s", "ttmp", "m0")):
            continue                                       # constant soffset: the compiler's recogniser handles it
        lo, hi = int(m.group(1)), int(m.group(2))
        waited, j = 0, i + 1
        while waited < min_wait_states and j < len(lines):
            nxt = lines[j].split("//")[0].rstrip()
            j += 1
            if not nxt.strip() or nxt.endswith(":"):
                continue
            n = _NOP.match(nxt)
            if n:
                waited += int(n.group(1)) + 1
                continue
            v = _VDST.match(nxt)
            if v and not v.group(1).startswith(_NO_VGPR_DST):
                a = int(v.group(2) if v.group(2) is not None else v.group(3))
                b = int(v.group(2) if v.group(2) is not None else v.group(4))
                if a <= hi and b >= lo:
                    found.append((symbol, line.strip(), nxt.strip()))
                    break
            if nxt.lstrip().startswith(("s_branch", "s_endpgm", "s_setpc")):
                break                                      # a taken branch costs more cycles than the hazard lasts
            waited += 1
    return found


def check_object(obj_path):
    """Raises RuntimeError when the object's device code contains the hazard."""
    bad = store_data_hazards(disassemble(obj_path))
    if bad:
        msg = "\n".join("  %s: %s  ->  %s" % b for b in bad[:8])
        raise RuntimeError("%s: %d wide buffer store(s) with an SGPR soffset are followed by a vector write of their data "
                           "(fence them: scheduling barriers + s_nop 2, see conv_h16.hip):\n%s" % (os.path.basename(obj_path), len(bad), msg))
