"""The documents keep the shape VERDICT r5 asked for: DESIGN.md is ONE front page about the current design (<= 250 lines,
lines <= 120 characters; the round-by-round log lives in docs/HISTORY.md), the drop-in module stays small, and every file a
document points to exists."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_md_is_a_front_page():
    lines = open(os.path.join(ROOT, "DESIGN.md"), encoding="utf-8").read().split("\n")
    assert len(lines) <= 250, len(lines)
    long = [(i + 1, len(l)) for i, l in enumerate(lines) if len(l) > 120]
    assert not long, long
    assert os.path.getsize(os.path.join(ROOT, "docs", "HISTORY.md")) > 100000          # (the log was moved, not dropped)


def test_the_drop_in_module_stays_small():
    n = sum(1 for _ in open(os.path.join(ROOT, "climate_toolbox_amd", "aggregations.py")))
    assert n <= 450, n
    from climate_toolbox_amd import aggregations as A
    for name in ("weighted_aggregate_grid_to_regions", "_reindex_spatial_data_to_regions", "_aggregate_reindexed_data_to_regions",
                 "prepare_spatial_weights_data", "prepare_weights", "PreparedWeights", "clear_caches", "HOST_DEVICES", "results_on_device",
                 "_PLAN_CACHE", "_TABLE_MEMO", "_resolve_cells", "_factorize_labels", "_backup_fill", "_prefer_dense", "_fingerprint"):
        assert hasattr(A, name), name


def test_files_named_in_the_documents_exist():
    pat = re.compile(r"((?:tools|profiles|tests|docs|oracle|include|climate_toolbox_amd)/[A-Za-z0-9_./\-]+\.(?:py|sh|cpp|hip|h|jsonl|json|csv|txt|md|npz|c|inc))")
    missing = []
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md", "profiles/README.md"):
        text = open(os.path.join(ROOT, doc), encoding="utf-8").read()
        for m in pat.finditer(text):
            p = m.group(1)
            if p == "tests/test_climate_toolbox.py":                 # the REFERENCE's test file (cited, not ours)
                continue
            if not (os.path.exists(os.path.join(ROOT, p)) or os.path.exists(os.path.join(ROOT, "climate_toolbox_amd", p))):
                missing.append((doc, p))
    assert not missing, missing
