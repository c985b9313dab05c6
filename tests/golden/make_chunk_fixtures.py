"""Copies the reference's legacy compressed forward-index blobs (data fixtures of FixedByteChunkSVForwardIndexTest /
VarByteChunkSVForwardIndexTest, backward-compatibility cases) into tests/golden/.  Run in the build container, where
/root/reference exists; the GPU box only reads the copies.

  fixedByteCompressed.v2       2000 doubles i + 100.2356, SNAPPY, writer version 2      (FixedByteChunkSVForwardIndexTest.java:352-357)
  fixedByteSVRDoubles.v1       10009 doubles i, writer version 1 (always SNAPPY)        (:343-347)
  varByteStringsCompressed.v2  1000 strings data[i % 4], SNAPPY, version 2              (VarByteChunkSVForwardIndexTest.java:153-159)
  varByteStrings.v1            1009 strings, version 1                                   (:143-148)
"""
import gzip
import os
import shutil

SRC = "/root/reference/pinot-segment-local/src/test/resources/data"
DST = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    for name in ("fixedByteCompressed.v2", "fixedByteSVRDoubles.v1", "varByteStringsCompressed.v2", "varByteStrings.v1"):
        with open(os.path.join(SRC, name), "rb") as f, gzip.GzipFile(os.path.join(DST, name + ".gz"), "wb", mtime=0) as g:
            shutil.copyfileobj(f, g)
        print(name, os.path.getsize(os.path.join(DST, name + ".gz")))
