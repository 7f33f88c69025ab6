"""profiles/hbm_traffic.json from the two rocprofv3 PMC summaries (FETCH_SIZE / WRITE_SIZE passes) of
tools/collect_evidence.sh, stamped with the digest of the kernel sources they were collected on (bench.py only reports
`roofline.traffic` when the stamp matches the build it is running).

    python tools/make_hbm_traffic.py gpurun_out/pmc_fetch_TAG.txt gpurun_out/pmc_write_TAG.txt TAG

HBM read bytes = 2 x FETCH_SIZE KiB (gfx950 correction, MI355X_MICROARCH.md HBM section); written bytes = WRITE_SIZE KiB."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_digest  # noqa: E402

CLASSES = [  # (substring of the demangled kernel name, bench.py kernel class)
    ("k_conv_ffn<15, false, true>", "k_conv_ffn+ffn_qkv"), ("k_conv_ffn<15, false, false>", "k_conv_ffn"),
    ("k_attn_out_glu", "k_attn_out_glu"), ("k_ffn_qkv", "k_ffn_qkv"), ("k_conv1", "k_conv1"),
    ("Conv2S", "k_gemm_stream<conv2>"), ("Dense", "k_gemm_stream<embed>"), ("k_ctc_head", "k_ctc_head"),
    ("k_attention", "k_attention"), ("k_out_glu", "k_out_glu"),
]


def parse(path, counter):
    out = {}
    for line in open(path):
        parts = [p.strip() for p in line.split("|")]
        if len(parts) == 5 and parts[1] == counter:
            for sub, cls in CLASSES:
                if sub in parts[0]:
                    n, avg = int(parts[2]), float(parts[3])
                    a = out.setdefault(cls, [0, 0.0])
                    a[0] += n
                    a[1] += n * avg
                    break
    return {k: v[1] / v[0] for k, v in out.items() if v[0]}


def main():
    fetch, write, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    f, w = parse(fetch, "FETCH_SIZE"), parse(write, "WRITE_SIZE")
    doc = {"_source": f"profiles/{tag}_pmc_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; read bytes = "
                      "2 x FETCH_SIZE KiB per the gfx950 note in MI355X_MICROARCH.md; includes Infinity-Cache hits)",
           "csrc_sha256": csrc_digest()}
    for cls in sorted(set(f) | set(w)):
        fk, wk = f.get(cls, 0.0), w.get(cls, 0.0)
        doc[cls] = {"fetch_kib": round(fk), "write_kib": round(wk), "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
    with open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w") as fh:
        json.dump(doc, fh, indent=2)
    print(json.dumps(doc, indent=2))


if __name__ == "__main__":
    main()
