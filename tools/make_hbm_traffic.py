"""profiles/hbm_traffic[_CFG].json from the two rocprofv3 PMC summaries (FETCH_SIZE / WRITE_SIZE passes) of
tools/collect_evidence.sh, stamped with the digest of the kernel sources they were collected on (bench.py only reports
`roofline.traffic` when the stamp matches the build it is running).

    python tools/make_hbm_traffic.py gpurun_out/pmc_fetch_TAG.txt gpurun_out/pmc_write_TAG.txt TAG [CFG [KERNEL_TRACE.txt]]

With the kernel-trace summary of the same evidence pass (tools/rocpd_summary.py output) every class also carries
`rocprof_avg_us` / `rocprof_calls`, which bench.py prints next to its own HIP-event figure (`avg_launch_ms_rocprof`).

Entries are keyed by bench.py's kernel classes (bench.class_of).  HBM read bytes = 2 x FETCH_SIZE KiB (gfx950
correction, MI355X_MICROARCH.md HBM section); written bytes = WRITE_SIZE KiB."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import class_of, csrc_digest  # noqa: E402


def short_name(full):
    """'void ppasr::k_conv_ffn<15, false, true>(float const*, ...' -> 'k_conv_ffn<15, false, true>' (the parameter list
    may be cut off in a summary: the name ends at the first '(' outside the template arguments)"""
    n = re.sub(r"^void ", "", full.strip())
    depth = 0
    for i, ch in enumerate(n):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            n = n[:i]
            break
    return n.replace("ppasr::", "").replace("(anonymous namespace)::", "")


def parse(path, counter):
    out = {}
    for line in open(path):
        parts = [p.strip() for p in line.split("|")]
        if len(parts) == 5 and parts[1] == counter and "ppasr::" in parts[0]:
            cls = class_of(short_name(parts[0]))
            n, avg = int(parts[2]), float(parts[3])
            a = out.setdefault(cls, [0, 0.0])
            a[0] += n
            a[1] += n * avg
    return {k: v[1] / v[0] for k, v in out.items() if v[0]}


def parse_instances(path, counter):
    """{class: {kernel instance: average counter value per dispatch}}"""
    out = {}
    for line in open(path):
        parts = [p.strip() for p in line.split("|")]
        if len(parts) == 5 and parts[1] == counter and "ppasr::" in parts[0]:
            n = short_name(parts[0])
            a = out.setdefault(class_of(n), {}).setdefault(n, [0, 0.0])
            a[0] += int(parts[2])
            a[1] += int(parts[2]) * float(parts[3])
    return {c: {n: v[1] / v[0] for n, v in d.items() if v[0]} for c, d in out.items()}


def parse_trace(path):
    """{class: (calls, avg_us)} from a '# name | calls | total_us | avg_us | pct' summary"""
    out = {}
    for line in open(path):
        parts = [p.strip() for p in line.split("|")]
        if len(parts) == 5 and "ppasr::" in parts[0] and not line.startswith("#"):
            cls = class_of(short_name(parts[0]))
            a = out.setdefault(cls, [0, 0.0])
            a[0] += int(parts[1])
            a[1] += float(parts[2])
    return {k: (v[0], v[1] / v[0]) for k, v in out.items() if v[0]}


def parse_trace_instances(path):
    """{class: {kernel instance: [calls, avg_us]}} -- bench.py weights a class's instances by ITS OWN launch mix (a trace
    that also holds the serial / fp16 x3 legs of the command has another mix of block forms per class)"""
    out = {}
    for line in open(path):
        parts = [p.strip() for p in line.split("|")]
        if len(parts) == 5 and "ppasr::" in parts[0] and not line.startswith("#"):
            n = short_name(parts[0])
            out.setdefault(class_of(n), {})[n] = [int(parts[1]), round(float(parts[3]), 3)]
    return out


def main():
    fetch, write, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    cfg = sys.argv[4] if len(sys.argv) > 4 else "cfg2"
    have_trace = len(sys.argv) > 5 and os.path.exists(sys.argv[5])
    trace = parse_trace(sys.argv[5]) if have_trace else {}
    inst = parse_trace_instances(sys.argv[5]) if have_trace else {}
    f, w = parse(fetch, "FETCH_SIZE"), parse(write, "WRITE_SIZE")
    fi, wi = parse_instances(fetch, "FETCH_SIZE"), parse_instances(write, "WRITE_SIZE")
    src = f"profiles/{tag}_pmc_hbm_traffic.txt" if cfg == "cfg2" else f"profiles/{tag}_{cfg}_pmc_hbm_traffic.txt"
    doc = {"_source": f"{src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; read bytes = "
                      "2 x FETCH_SIZE KiB per the gfx950 note in MI355X_MICROARCH.md; includes Infinity-Cache hits)",
           "csrc_sha256": csrc_digest(), "config": cfg}
    for cls in sorted(set(f) | set(w)):
        fk, wk = f.get(cls, 0.0), w.get(cls, 0.0)
        doc[cls] = {"fetch_kib": round(fk), "write_kib": round(wk), "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
        names = sorted(set(fi.get(cls, {})) | set(wi.get(cls, {})))
        if len(names) > 1:  # several instances (block forms): per-instance bytes, weighted by the line's launch mix in bench.py
            doc[cls]["hbm_bytes_per_launch_instances"] = {
                n: int((2 * fi.get(cls, {}).get(n, 0.0) + wi.get(cls, {}).get(n, 0.0)) * 1024) for n in names}
        if cls in trace:
            doc[cls].update(rocprof_calls=trace[cls][0], rocprof_avg_us=round(trace[cls][1], 3), rocprof_instances=inst.get(cls, {}))
    name = "hbm_traffic.json" if cfg == "cfg2" else f"hbm_traffic_{cfg}.json"
    with open(os.path.join(ROOT, "profiles", name), "w") as fh:
        json.dump(doc, fh, indent=2)
    print(json.dumps(doc, indent=2))


if __name__ == "__main__":
    main()
