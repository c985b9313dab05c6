"""The plug-in's Java sources (integration/java) against the reference's sources: tools/java_api_check.py, the compile-substitute of an
image without a JDK (VERDICT r3 #3b).  Runs where /root/reference exists (this container); skipped on the GPU box.

Besides "the real sources resolve", the checker itself is tested: seeded defects of the kinds a javac run would flag — a misspelt method,
a constructor of the wrong arity, an argument of the wrong type, a `case` label that is no constant of the enum, an abstract method left
unimplemented, an import of a class that does not exist — must each be reported.
"""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
SOURCES = os.path.join(ROOT, "integration", "java")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "pinot-core")), reason="needs the reference sources")


def _checker():
    spec = importlib.util.spec_from_file_location("java_api_check", os.path.join(ROOT, "tools", "java_api_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_plugin_sources_resolve_against_the_reference():
    chk, files = _checker().run(REFERENCE, SOURCES)
    assert len(files) >= 8
    assert chk.errors == []
    # the check is not vacuous: hundreds of uses were actually looked up in the reference's sources
    assert chk.checked >= 600
    assert chk.untyped <= 10


MUTATIONS = [
    ("NativeQuery.java", "q.getNumGroupsLimit()", "q.getNumGroupLimit()", "no method `getNumGroupLimit`"),
    ("GpuGroupByOperator.java", "new AggregationResultsBlock(functions, results, _queryContext)", "new AggregationResultsBlock(functions, results)",
     "no constructor matches"),
    ("GpuResultObjects.java", "dictionary.getIntValue(dictIds[at + i])", "dictionary.getIntValue(\"seven\")", "no overload matches (types)"),
    ("NativeQuery.java", "case COUNTMV: return 8;", "case COUNTMVX: return 8;", "not a constant of enum"),
    ("GpuGroupByOperator.java", "public int getNumKeys() {", "public int getNumberOfKeys() {", "does not implement"),
    ("GpuGroupByOperator.java", "import org.apache.pinot.core.operator.blocks.results.GroupByResultsBlock;",
     "import org.apache.pinot.core.operator.blocks.GroupByResultsBlock;", "not found in the reference"),
    ("GpuSegmentRegistry.java", "md.getTotalNumberOfEntries()", "md.getTotalNumberOfEntries(1)", "no overload matches (arity)"),
    ("GpuInstancePlanMaker.java", "QueryContextUtils.isAggregationQuery(queryContext)", "QueryContextUtils.isAggregationQuery(segment)",
     "no overload matches (types)"),
    ("GpuGroupByOperator.java", "block.setNumGroupsLimitReached(stats[4] != 0);", "block.setNumGroupsLimitReached(stats[4]);",
     "no overload matches (types)"),
    ("GpuGroupByCombineOperator.java", "super(operators, queryContext, executorService);", "super(operators, queryContext);", "no constructor matches"),
    ("GpuGroupByCombineOperator.java", "return super.getNextBlock();", "return super.getNextResultsBlock();", "no method `getNextResultsBlock`"),
]


@pytest.mark.parametrize("file,old,new,expect", MUTATIONS, ids=[m[3] + ":" + m[0] for m in MUTATIONS])
def test_seeded_defects_are_reported(tmp_path, file, old, new, expect):
    dst = tmp_path / "java"
    shutil.copytree(SOURCES, dst)
    path = None
    for dp, _, fn in os.walk(dst):
        if file in fn:
            path = os.path.join(dp, file)
    text = open(path).read()
    assert old in text, "the mutation's anchor left the sources: update the test"
    open(path, "w").write(text.replace(old, new, 1))
    chk, _ = _checker().run(REFERENCE, str(dst))
    assert any(expect in e for e in chk.errors), chk.errors
