"""Generates tests/golden/fast_filtered_count_cases.json from the reference's own test (run in the container that holds /root/reference):
FastFilteredCountTest.java's data provider (:147-308) — `select count(*) ... where ...` strings with their expected counts over the formulaic
table of :104-113 (1 000 records: class = i % 8, sorted = i, intRangeCol = 1000 - i).  The Java string concatenations are evaluated as Python
expressions; the TEXT_MATCH / JSON_MATCH cases (indexes outside the path) are dropped."""
import json
import os
import re

SRC = "/root/reference/pinot-core/src/test/java/org/apache/pinot/queries/FastFilteredCountTest.java"
lines = open(SRC).read().split("\n")
start = next(i for i, l in enumerate(lines) if "return new Object[][] {" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith("    };"))
body = "\n".join(lines[start + 1:end])
class JInt(int):
    """a Java int: concatenates with strings, stays a JInt under arithmetic (integer division)"""
    def __add__(self, o): return (str(int(self)) + o) if isinstance(o, str) else JInt(int(self) + int(o))
    def __radd__(self, o): return (o + str(int(self))) if isinstance(o, str) else JInt(int(o) + int(self))
    def __sub__(self, o): return JInt(int(self) - int(o))
    def __rsub__(self, o): return JInt(int(o) - int(self))
    def __mul__(self, o): return JInt(int(self) * int(o))
    def __rmul__(self, o): return JInt(int(o) * int(self))
    def __floordiv__(self, o): return JInt(int(self) // int(o))
    def __rfloordiv__(self, o): return JInt(int(o) // int(self))


env = {"NUM_RECORDS": JInt(1000), "BUCKET_SIZE": JInt(8), "RAW_TABLE_NAME": "testTable", "SORTED_COLUMN": "sorted", "CLASSIFICATION_COLUMN": "class",
       "TEXT_COLUMN": "textCol", "JSON_COLUMN": "jsonCol", "INT_RANGE_COLUMN": "intRangeCol"}
env["bucketCount"] = env["NUM_RECORDS"] // env["BUCKET_SIZE"]
env["bucketCountComplement"] = env["NUM_RECORDS"] - env["bucketCount"]
env["min"], env["max"] = JInt(20), env["NUM_RECORDS"] - 20
env["allBuckets"] = "(" + ", ".join(str(v) for v in range(8)) + ")"
env["twoBuckets"] = "(0, 7)"
# one case = "{" expr "," expr "}," possibly over several lines
cases = []
for m in re.finditer(r"\{(\"select.*?)\}\s*,?\s*(?=\{\"select|\Z)", body, re.S):
    text = m.group(1).replace("\n", " ")
    text = text.replace(" / ", " // ")
    text = re.sub(r"(?<![\w\"])(\d+)(?![\w\"])", r"JInt(\1)", text)   # bare integer literals
    query_src, expected_src = text.rsplit(",", 1)
    query = eval(query_src, dict(env, JInt=JInt))        # noqa: S307 (the reference's own test source)
    expected = eval(expected_src, dict(env, JInt=JInt))  # noqa: S307
    if "text_match" in query.lower() or "json_match" in query.lower():
        continue
    cases.append({"query": query, "expected": int(expected)})
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "fast_filtered_count_cases.json")
json.dump({"source": "pinot-core/src/test/java/org/apache/pinot/queries/FastFilteredCountTest.java:104-113,147-308", "cases": cases}, open(out, "w"), indent=1)
print(len(cases), "cases ->", out)
