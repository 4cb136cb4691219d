#!/usr/bin/env python3
"""Static instruction budget of a plan-specialised code object (tools only): per kernel the instruction count by issue class,
code bytes and register use, and — for a build with -DMRX_PROFILE_PHASES — the static instruction count of every stretch of code
between two phase markers (s_memtime), in layout order.

    python tools/isa_budget.py maro_amd/csrc/spec_cache/<key>.hsaco [kernel name ...]        (works without a GPU)
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def classify(op: str) -> str:
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime", "s_dcache", "s_store", "s_atomic")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_endpgm", "s_sethalt", "s_setprio", "s_code_end")):
        return "wait/ctl"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def disassemble(path: str) -> str:
    with tempfile.TemporaryDirectory() as tmp:
        elf = os.path.join(tmp, "co.elf")
        head = open(path, "rb").read(24)
        if head.startswith(b"__CLANG_OFFLOAD_BUNDLE__"):
            subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={path}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                   f"--output={elf}"])
        else:
            elf = path
        return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", elf], capture_output=True, text=True, check=True).stdout


def kernels(dis: str) -> dict:
    out, cur = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        m = re.match(r"^\s+(\S+)\s.*//\s*([0-9A-Fa-f]+):((?:\s+[0-9A-Fa-f]{8})+)", line)
        if cur is not None and m:
            cur.append((m.group(1), 4 * len(m.group(3).split())))
    return out


def main():
    path, want = sys.argv[1], sys.argv[2:]
    ks = kernels(disassemble(path))
    classes = ["valu", "salu", "lds", "vmem", "smem", "branch", "wait/ctl", "mfma", "other"]
    print(f"## {os.path.basename(path)}\n")
    print("| kernel | instructions | " + " | ".join(classes) + " | code bytes |")
    print("|---|---|" + "---|" * (len(classes) + 1))
    for name, ins in ks.items():
        if want and name not in want:
            continue
        real = [(op, b) for op, b in ins if op != "s_code_end"]
        cnt = {c: 0 for c in classes}
        for op, _ in real:
            cnt[classify(op)] += 1
        print(f"| `{name}` | {len(real)} | " + " | ".join(str(cnt[c]) for c in classes) + f" | {sum(b for _, b in real)} |")
    for name, ins in ks.items():
        if (want and name not in want) or not any(op == "s_memtime" for op, _ in ins):
            continue
        print(f"\n### `{name}`: static instructions between consecutive phase markers (s_memtime), in layout order\n")
        print("| stretch | instructions | valu | salu | lds | vmem | smem | branch | wait/ctl |")
        print("|---|---|---|---|---|---|---|---|---|")
        seg, k = {c: 0 for c in classes}, 0
        for op, _ in ins + [("s_memtime", 0)]:
            if op == "s_memtime":
                n = sum(seg.values())
                if n:
                    print(f"| {k} | {n} | " + " | ".join(str(seg[c]) for c in ("valu", "salu", "lds", "vmem", "smem", "branch", "wait/ctl")) + " |")
                seg, k = {c: 0 for c in classes}, k + 1
            elif op != "s_code_end":
                seg[classify(op)] += 1


if __name__ == "__main__":
    main()
