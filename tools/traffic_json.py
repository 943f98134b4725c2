#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of `tools/fcn_bench.py --batch 8 --iters 1` into the
profiles/rNN_conv_hbm_traffic.json that bench.py reads.   python tools/traffic_json.py <fetch.csv> <write.csv> <out.json>"""
import json, re, sys
import pandas as pd

f, w = pd.read_csv(sys.argv[1]), pd.read_csv(sys.argv[2])


def short(n):
    mm = re.search(r'k_convI\w+?Li(\d+)ELi(\d+)', n)
    if mm:
        return 'k_conv<%s,%s>' % (mm.group(1), mm.group(2))
    mb = re.search(r'k_conv_blockILi(\d+)ELb(\d)', n)
    if mb:
        return 'k_conv_block<%s,%s>' % (mb.group(1), 'chain' if mb.group(2) == '1' else 'plain')
    m = re.search(r'(k_[a-z_0-9]+)', n)
    return m.group(1) if m else n[:40]


f['k'], w['k'] = f.Kernel_Name.map(short), w.Kernel_Name.map(short)
t = f.groupby('k').agg(dispatches=('Dispatch_Id', 'nunique'), fetch_kb=('Counter_Value', 'sum')).join(
    w.groupby('k').agg(write_kb=('Counter_Value', 'sum')))
conv = t[t.index.str.startswith('k_conv') | (t.index == 'k_tail')]
n_fw = int(t.loc['k_tail', 'dispatches'])
fetch, write = conv.fetch_kb.sum(), conv.write_kb.sum()
total = (2 * fetch + write) * 1024
out = {
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/fcn_bench.py --batch 8 --iters 1",
    "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts 64 of every 128 B of wide coalesced reads; both counters are in KB)",
    "conv_family": {"kernels": ", ".join(sorted(conv.index)), "launches": int(conv.dispatches.sum()), "forward_batches": n_fw,
                    "fetch_size_kb": float(fetch), "write_size_kb": float(write), "hbm_bytes_per_step_8frames_3nets": total / n_fw,
                    "hbm_bytes_per_network_forward": total / n_fw / 24, "hbm_bytes_per_launch": total / float(conv.dispatches.sum())},
    "per_kernel": [{"kernel": k, "dispatches": int(r.dispatches), "fetch_size_kb": float(r.fetch_kb), "write_size_kb": float(r.write_kb)}
                   for k, r in t.iterrows()],
}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print("%.3f GB per network forward" % (total / n_fw / 24 / 1e9))
