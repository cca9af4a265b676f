"""ISA audit of the built product library (CPU only, llvm-objdump): per kernel, the compiler-inserted serialisations that cost round 3
its last 8 % before they were found by hand:

  * `s_waitcnt vmcnt(0)` immediately in front of a `ds_read`: the waitcnt pass decided that an LDS read may depend on an LDS-DMA
    instruction still in flight and waits for ALL of them, i.e. for the prefetch that was just issued;
  * `s_waitcnt vmcnt(0)` immediately after a vector-memory load: a load waited for on the spot (inside a loop: a chain of latencies);
  * scratch (spill) instructions.

usage: python tools/isa_audit.py [substring of the kernel name ...]     (default: every kernel, one summary line each)
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "diffpir_amd", "csrc", "libdiffpir_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def device_code_objects(blob):
    pos = 0
    while True:
        i = blob.find(MAGIC, pos)
        if i < 0:
            return
        n = struct.unpack_from("<Q", blob, i + 24)[0]
        p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            p += 24
            triple = blob[p:p + tl]
            p += tl
            if b"gfx950" in triple and size:
                yield blob[i + off:i + off + size]
        pos = i + len(MAGIC)


def disassemble(so=SO):
    """{kernel symbol: [mnemonic + operands, ...]} for every gfx950 code object bundled in the library."""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for k, obj in enumerate(device_code_objects(open(so, "rb").read())):
            f = os.path.join(tmp, f"o{k}.elf")
            open(f, "wb").write(obj)
            asm = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f], capture_output=True, text=True, check=True).stdout
            sym = None
            for line in asm.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
                if m:
                    sym = m.group(1)
                    out[sym] = []
                    continue
                if sym is None:
                    continue
                ins = line.split("//")[0].strip()
                if ins and not ins.startswith("."):
                    out[sym].append(ins)
    return out


def audit(instrs):
    r = {"instructions": len(instrs), "mfma": 0, "lds_dma": 0, "vmcnt0": 0, "vmcnt0_before_ds_read": 0, "vmcnt0_after_load": 0, "scratch": 0}
    is_load = lambda s: re.match(r"^(global_load|buffer_load|flat_load)", s) and " lds" not in s
    for i, s in enumerate(instrs):
        if s.startswith("v_mfma"):
            r["mfma"] += 1
        if re.match(r"^(global_load|buffer_load).* lds", s) or s.startswith("global_load_lds"):
            r["lds_dma"] += 1
        if s.startswith("scratch_"):
            r["scratch"] += 1
        if re.match(r"^s_waitcnt\b.*vmcnt\(0\)", s):
            r["vmcnt0"] += 1
            nxt = instrs[i + 1] if i + 1 < len(instrs) else ""
            prv = instrs[i - 1] if i else ""
            if nxt.startswith("ds_read"):
                r["vmcnt0_before_ds_read"] += 1
            if is_load(prv):
                r["vmcnt0_after_load"] += 1
    return r


def main():
    pats = sys.argv[1:]
    for sym, ins in sorted(disassemble().items()):
        if pats and not any(p in sym for p in pats):
            continue
        r = audit(ins)
        if r["instructions"] < 8:
            continue
        print(f"{sym[:96]:96s} n={r['instructions']:6d} mfma={r['mfma']:4d} lds_dma={r['lds_dma']:3d} vmcnt0={r['vmcnt0']:3d} "
              f"before_ds_read={r['vmcnt0_before_ds_read']:2d} after_load={r['vmcnt0_after_load']:2d} scratch={r['scratch']:3d}")


if __name__ == "__main__":
    main()
