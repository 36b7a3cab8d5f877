#!/usr/bin/env python3
"""rocprofv3 PC-sampling CSVs of a directory -> {column names, first rows, samples per
(instruction / pc) key}: the per-instruction histogram of where a wavefront's time goes."""
import collections
import csv
import glob
import json
import os
import sys
src, dst = sys.argv[1], sys.argv[2]
res = {"files": [], "hist": {}}
for f in glob.glob(os.path.join(src, "**", "*.csv"), recursive=True):
    size = os.path.getsize(f)
    info = {"file": os.path.relpath(f, src), "bytes": size}
    res["files"].append(info)
    if "pc_sampling" not in os.path.basename(f):
        if size < 20000:
            info["text"] = open(f).read()
        continue
    with open(f, newline="") as fh:
        rd = csv.reader(fh)
        head = next(rd, None)
        info["columns"] = head
        rows = []
        cnt = collections.Counter()
        extra = collections.defaultdict(collections.Counter)
        n = 0
        # key: every column that names the instruction / its address; other low-cardinality
        # columns (stall reason, instruction type, issued ...) as per-key sub-histograms
        keycols = [i for i, c in enumerate(head) if any(k in c.lower() for k in ("instruction", "pc", "offset")) and "comment" not in c.lower()]
        skip = [i for i, c in enumerate(head) if any(k in c.lower() for k in ("timestamp", "correlation", "exec", "dispatch", "wave", "chiplet", "hw_id", "workgroup"))]
        for row in rd:
            n += 1
            if len(rows) < 5:
                rows.append(row)
            key = " | ".join(row[i] for i in keycols)
            cnt[key] += 1
            for i, c in enumerate(head):
                if i not in keycols and i not in skip:
                    extra[key][c + "=" + row[i]] += 1
        info["rows"] = n
        info["first_rows"] = rows
        res["hist"][info["file"]] = [{"key": k, "n": v, "extra": dict(extra[k].most_common(12))} for k, v in cnt.most_common(6000)]
json.dump(res, open(dst, "w"))
print("aggregated", [(i["file"], i.get("rows")) for i in res["files"]])
