# Kernel trace of KV-cached decode steps (B = 1): per-kernel durations and inter-kernel gaps of one step in each multi-launch mode.
export TMPDIR=/tmp
rm -rf /tmp/dp
rocprofv3 --kernel-trace --output-format csv -d /tmp/dp -o dp -- python scripts/decode_step_probe.py ${1:-1} > /tmp/dp.log 2>&1
f=$(find /tmp/dp -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); return re.sub(r"\(.*$", "", n)[:44]
names = [short(r["Kernel_Name"]) for r in rows]
emb = [i for i, n in enumerate(names) if n.startswith("embedding_fwd")]
# steps = spans between consecutive embedding kernels; classify by content
steps = collections.defaultdict(list)
for a, b in zip(emb[:-1], emb[1:]):
    span = rows[a:b]
    if not (60 < len(span) < 200):
        continue
    kind = "default (decode_proj)" if any("decode_proj" in short(r["Kernel_Name"]) for r in span) else "mode 2 (LN + skinny GEMM)"
    steps[kind].append(span)
for kind, lst in steps.items():
    lst = lst[len(lst) // 2: len(lst) // 2 + 40]
    dur = collections.defaultdict(list); gaps = []; tot = []
    for span in lst:
        prev = None
        for r in span:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            dur[short(r["Kernel_Name"])].append((e - s) / 1e3)
            if prev is not None: gaps.append((s - prev) / 1e3)
            prev = e
        tot.append((int(span[-1]["End_Timestamp"]) - int(span[0]["Start_Timestamp"])) / 1e3)
    n = len(lst)
    print(f"== {kind}: {len(lst[0])} kernels per step, span {sum(tot)/n:.0f} us, kernel time {sum(sum(v) for v in dur.values())/n:.0f} us, gaps {sum(gaps)/n:.0f} us")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print(f"   {k:44s} x{len(v)//n:3d}  avg {sum(v)/len(v):6.2f} us  total/step {sum(v)/n:7.1f} us")
PY
