"""Generates tests/golden/labels.json with the reference's own `prepare_train_dataset` (run_distillation.py:1167-1229,
exec'd from /root/reference with a stub tokenizer that returns pre-tokenised ids).  Run in the build container:
python oracle/gen_golden_labels.py"""
import json, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from tests.test_labels import make_batch, reference_labels

rng = np.random.default_rng(123)
cases = []
for trial in range(12):
    toks, prevs = make_batch(rng, int(rng.integers(1, 7)), with_column=trial % 2 == 0)
    tp, cp = [(0.2, 0.2), (1.0, 1.0), (0.0, 1.0), (0.5, 0.9)][trial % 4]
    cases.append({"seed": 500 + trial, "tp": tp, "cp": cp, "tokens": toks, "prevs": prevs,
                  "labels": reference_labels(toks, prevs, 500 + trial, tp, cp)})
path = os.path.join(ROOT, "tests", "golden", "labels.json")
json.dump(cases, open(path, "w"))
print("wrote", path, len(cases))
