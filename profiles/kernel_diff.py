"""Which device kernels differ between two builds of libmpmhip.so?
    python profiles/kernel_diff.py old/libmpmhip.so new/libmpmhip.so
Extracts the gfx950 code object of each library (llvm-objdump --offloading), disassembles it and compares every kernel's
instruction stream.  Used to check that an edit meant to be a no-op for the tuned kernels really is one before spending GPU
time on it — it is not always: removing ONE unused variable from k_g2p changed its register allocation (5610 -> 5611
instructions), so the variable stayed (k_g2p.h)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(lib):
    d = tempfile.mkdtemp()
    try:
        shutil.copy(lib, os.path.join(d, "lib.so"))
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=d, check=True, stdout=subprocess.DEVNULL)
        co = [f for f in os.listdir(d) if f.endswith("gfx950")][0]
        txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], cwd=d, check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(d)
    parts = re.split(r"^[0-9a-f]+ <([^>]+)>:$", txt, flags=re.M)
    return {parts[i]: [re.sub(r"\s*//.*$", "", ln.strip()) for ln in parts[i + 1].splitlines() if ln.strip() and not ln.strip().startswith("//")]
            for i in range(1, len(parts), 2)}


def kernel_stats(lib):
    """per kernel (mangled name): instruction count of the disassembly + the resource usage the code object's metadata
    records (vgpr / agpr / sgpr counts, spills, scratch bytes, static LDS bytes)"""
    d = tempfile.mkdtemp()
    try:
        shutil.copy(lib, os.path.join(d, "lib.so"))
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=d, check=True, stdout=subprocess.DEVNULL)
        co = [f for f in os.listdir(d) if f.endswith("gfx950")][0]
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], cwd=d, check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(d)
    out, cur = {}, None
    keys = {".vgpr_count": "vgpr", ".agpr_count": "agpr", ".sgpr_count": "sgpr", ".vgpr_spill_count": "vgpr_spill",
            ".sgpr_spill_count": "sgpr_spill", ".private_segment_fixed_size": "scratch_bytes", ".group_segment_fixed_size": "lds_bytes"}
    for blk in re.split(r"^\s+- \.", notes, flags=re.M):  # one metadata map per kernel
        m = re.search(r"^\s*\.?name:\s+(\S+)$", blk, flags=re.M)
        sym = re.search(r"\.symbol:\s+(\S+)\.kd", blk)
        if not sym:
            continue
        cur = {}
        for k, short in keys.items():
            mm = re.search(r"%s:\s+(\d+)" % re.escape(k), blk)
            if mm:
                cur[short] = int(mm.group(1))
        out[sym.group(1)] = cur
    for name, ins in kernels(lib).items():
        if name in out:
            out[name]["instructions"] = len(ins)
    return out


def base_name(mangled):
    """'k_g2p' of _ZN3mpm5k_g2pILi256E...: the kernel's name without namespace and template arguments (the key of
    profiles/traffic_*.json, where make_traffic.py cuts rocprofv3's demangled names the same way)"""
    m = re.match(r"^_ZN3mpm(\d+)", mangled)
    if not m:
        return None
    n, at = int(m.group(1)), m.end()
    return mangled[at:at + n]


def kernel_hashes(lib):
    """{base name: sha256 over the instruction streams of ALL its instantiations} — the identity of a kernel's code as a PMC summary
    or a bench line can record it: any edit that reaches one instantiation changes the hash (conservative on purpose)"""
    import hashlib
    acc = {}
    for name, ins in sorted(kernels(lib).items()):
        b = base_name(name)
        if b:
            acc.setdefault(b, hashlib.sha256()).update(("\n".join([name] + ins) + "\n").encode())
    return {k: h.hexdigest()[:16] for k, h in acc.items()}


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--hashes":
        import json
        print(json.dumps(kernel_hashes(sys.argv[2]), indent=1, sort_keys=True))
        sys.exit(0)
    if len(sys.argv) == 3 and sys.argv[1] == "--stats":
        import json
        print(json.dumps(kernel_stats(sys.argv[2]), indent=1, sort_keys=True))
        sys.exit(0)
    a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
    names = sorted(set(a) | set(b))
    diff = [k for k in names if a.get(k) != b.get(k)]
    print("%d kernels, %d differ" % (len(names), len(diff)))
    for k in diff:
        print("  %-90s %s -> %s instructions" % (k[:90], len(a[k]) if k in a else "-", len(b[k]) if k in b else "-"))
