"""Copies the reference's v1-layout segment fixtures (written by Pinot itself in 2016, 5 docs each, one file per index:
`<col>.dict`, `<col>.sv.unsorted.fwd`, `metadata.properties`) into tests/golden/: paddingNull (dictionary strings padded with
\\0), paddingPercent ('%') and paddingOld (no padding property: the legacy '%').  Run in the build container."""
import os
import shutil

SRC = "/root/reference/pinot-core/src/test/resources/data"
DST = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    for name in ("paddingNull", "paddingPercent", "paddingOld"):
        shutil.copyfile(os.path.join(SRC, name + ".tar.gz"), os.path.join(DST, name + ".tar.gz"))
        print(name, os.path.getsize(os.path.join(DST, name + ".tar.gz")))
