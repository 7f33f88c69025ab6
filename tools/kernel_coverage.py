"""Which compiled kernels does the GPU test suite launch?

    python tools/kernel_coverage.py --list            # build box: every __global__ instantiation of csrc/*.hip ->
                                                      # tools/_ts/kernels_compiled.txt (travels with gpurun) + profiles/
    PPASR_KCOV=gpurun_out/kcov.tsv python -m pytest tests -m gpu -q     # GPU box: tests/conftest.py records each test's kernels
    python tools/kernel_coverage.py --report gpurun_out/kcov.tsv        # compare with the compiled list
    python tools/kernel_coverage.py --report-db DB...                   # the same from rocprofv3 rocpd databases

The compiled list comes from the device assembly (`hipcc -S --cuda-device-only`: one `.amdhsa_kernel` per kernel), the
launched set from the library's own per-kernel profile (ppasr_kprof_*) around every GPU test."""
import glob
import os
import re
import sqlite3
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIST = os.path.join(ROOT, "tools", "_ts", "kernels_compiled.txt")
CXXFILT = "c++filt"


def short(full):
    """demangled name without return type, namespaces and the parameter list"""
    n = re.sub(r"^void ", "", full.strip()).replace("(anonymous namespace)::", "")
    m = re.match(r"_ZN(?:\d+[A-Za-z_]\w*?)*?(\d+)", n)
    if n.startswith("_Z"):  # a mangling this c++filt does not know (_Float16 parameters): keep the last name segment
        segs = re.findall(r"(\d+)([A-Za-z_]\w*)", n.split("E", 1)[0] + "E")
        i, out = 3 if n.startswith("_ZN") else 2, []
        while i < len(n) and n[i].isdigit():
            j = i
            while n[j].isdigit():
                j += 1
            ln = int(n[i:j])
            out.append(n[j:j + ln])
            i = j + ln
        return out[-1] if out else n
    depth = 0
    for i, ch in enumerate(n):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            n = n[:i]
            break
    return n.replace("ppasr::", "").replace("(anonymous namespace)::", "").strip()


def compiled():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    srcs = sorted(glob.glob(os.path.join(entry.CSRC, "*.hip")))

    def one(src):
        out = subprocess.run(["hipcc"] + entry.FLAGS + ["-S", "--cuda-device-only", src, "-o", "-"], check=True,
                             capture_output=True, text=True).stdout
        return [(os.path.basename(src), m) for m in re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", out, re.M)]

    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as ex:
        pairs = [p for ps in ex.map(one, srcs) for p in ps]
    dem = subprocess.run([CXXFILT], input="\n".join(m for _f, m in pairs), capture_output=True, text=True, check=True).stdout.split("\n")
    return sorted({(short(d), f) for (f, _m), d in zip(pairs, dem) if d.strip()})


def launched(dbs):
    names = {}
    for db in dbs:
        try:
            for name, calls in sqlite3.connect(db).execute("select name,total_calls from top_kernels"):
                names[short(name)] = names.get(short(name), 0) + calls
        except sqlite3.Error as e:
            print(f"# {db}: {e}")
    return names


def launched_tsv(paths):
    names, tests = {}, {}
    for p in paths:
        for ln in open(p):
            k, n, t = ln.rstrip("\n").split("\t")
            k = short(k)  # (a name the runtime's demangler could not resolve arrives mangled)
            names[k] = names.get(k, 0) + int(n)
            tests.setdefault(k, set()).add(t.split("::")[0])
    return names, tests


def report(seen, source):
    comp = [ln.rstrip("\n").split("\t") for ln in open(LIST)]
    missing = [(k, f) for k, f in comp if k not in seen]
    print(f"# {len(comp)} kernels compiled into libppasr_hip.so, {len(comp) - len(missing)} launched by `pytest -m gpu` "
          f"({source}, {sum(seen.values())} dispatches)")
    print("# never launched by the GPU suite:")
    for k, f in missing:
        print(f"{k}\t{f}")
    extra = sorted(k for k in seen if k not in {c for c, _f in comp})
    if extra:
        print("# launched but not in the compiled list:", ", ".join(extra))
    return missing


if __name__ == "__main__":
    if "--list" in sys.argv:
        os.makedirs(os.path.dirname(LIST), exist_ok=True)
        ks = compiled()
        with open(LIST, "w") as f:
            f.writelines(f"{k}\t{src}\n" for k, src in ks)
        print(f"{len(ks)} kernels -> {LIST}")
    elif "--report-db" in sys.argv:
        dbs = [a for a in sys.argv[1:] if not a.startswith("--")]
        report(launched(dbs), f"{len(dbs)} rocprofv3 kernel-trace database(s)")
    else:
        files = [a for a in sys.argv[1:] if not a.startswith("--")]
        report(launched_tsv(files)[0], "ppasr_kprof around every test")
