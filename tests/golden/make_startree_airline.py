"""Extracts the reference's own star-tree fixture into tests/golden/startree_airline/ (run in the build container, where
/root/reference exists; the GPU box only sees the committed output).

Source (data, not code): pinot-segment-local/src/test/resources/data/startree/segment/{star_tree_index,
star_tree_index_map, metadata.properties} — the segment StarTreeIndexSeparatorTest.java:43-80 loads: a star-tree built by
the reference over airlineStats (313 docs → 1004 star-tree docs; split order AirlineID, Origin, Dest; function-column
pairs count__*, max__ArrDelay; maxLeafRecords 10).  `star_tree_index` is copied byte for byte; of the 1.3 k-line
metadata.properties only the keys the path reads are kept, as JSON.
"""
import json
import os
import re
import shutil

SRC = "/root/reference/pinot-segment-local/src/test/resources/data/startree/segment"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "startree_airline")


def properties(path):
    out = {}
    for line in open(path, encoding="utf-8"):
        line = line.strip()
        if not line or line.startswith("#") or "=" not in line:
            continue
        k, v = line.split("=", 1)
        out.setdefault(k.strip(), []).append(v.strip())
    return out


def main():
    os.makedirs(DST, exist_ok=True)
    shutil.copyfile(os.path.join(SRC, "star_tree_index"), os.path.join(DST, "star_tree_index"))
    os.chmod(os.path.join(DST, "star_tree_index"), 0o644)
    md = properties(os.path.join(SRC, "metadata.properties"))
    imap = properties(os.path.join(SRC, "star_tree_index_map"))
    dims = md["startree.v2.0.split.order"]
    meta = {
        "segment_name": md["segment.name"][0],
        "segment_total_docs": int(md["segment.total.docs"][0]),
        "star_tree_count": int(md["startree.v2.count"][0]),
        "total_docs": int(md["startree.v2.0.total.docs"][0]),
        "split_order": dims,
        "function_column_pairs": md["startree.v2.0.function.column.pairs"],
        "max_leaf_records": int(md["startree.v2.0.max.leaf.records"][0]),
        "columns": {c: {"cardinality": int(md[f"column.{c}.cardinality"][0]),
                        "bitsPerElement": int(md[f"column.{c}.bitsPerElement"][0]),
                        "dataType": md[f"column.{c}.dataType"][0],
                        "minValue": md[f"column.{c}.minValue"][0], "maxValue": md[f"column.{c}.maxValue"][0]}
                    for c in dims + ["ArrDelay"]},
        # star_tree_index_map: "<tree>.<column>.<INDEX>.<OFFSET|SIZE> = n" (StarTreeIndexMapUtils.java)
        "index_map": {k: int(v[0]) for k, v in imap.items() if re.match(r"^\d+\.", k)},
    }
    with open(os.path.join(DST, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    # the whole segment's index_map and per-column metadata (what pinot_amd/segment_dir.py parses), comments dropped
    keep = ("cardinality", "totalDocs", "dataType", "bitsPerElement", "lengthOfEachEntry", "isSorted", "hasDictionary",
            "isSingleValues", "maxNumberOfMultiValues", "totalNumberOfEntries")
    seg_meta = {
        "segment.total.docs": md["segment.total.docs"][0], "segment.name": md["segment.name"][0],
        "segment.index.version": md["segment.index.version"][0],
        "segment.padding.character": md["segment.padding.character"][0],
        "index_map": {k: v[0] for k, v in properties(os.path.join(SRC, "index_map")).items()},
        "columns": {k: v[0] for k, v in md.items() if k.startswith("column.") and k.rsplit(".", 1)[1] in keep},
        "startree": {k: v for k, v in md.items() if k.startswith("startree.")},
    }
    with open(os.path.join(DST, "segment_meta.json"), "w") as f:
        json.dump(seg_meta, f, indent=0, sort_keys=True)
    print(json.dumps(meta, indent=1)[:600])


if __name__ == "__main__":
    main()
