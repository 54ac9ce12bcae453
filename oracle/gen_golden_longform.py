"""Generates tests/golden/longform.json: known answers for the long-form scheduler's integer host logic, produced by
the `transformers` functions the reference's pipeline path runs (run_eval.py:566-576 -> ASR pipeline):
  chunk_iter                      (TF:pipelines/automatic_speech_recognition.py:61-84)
  _find_longest_common_sequence   (TF:models/whisper/tokenization_whisper.py)
Run in the build container: python oracle/gen_golden_longform.py"""
import json, os
import numpy as np
from transformers.pipelines.automatic_speech_recognition import chunk_iter
from transformers.models.whisper.tokenization_whisper import _find_longest_common_sequence


class _FE:
    sampling_rate = 16000

    def __call__(self, chunk, **kw):
        return {}


def spans(n, chunk_len, sl, sr):
    out, start, step = [], 0, chunk_len - sl - sr
    # chunk_iter does not report the start offset: recover it from its iteration rule (start advances by `step`)
    it = chunk_iter(np.zeros(n, np.float32), _FE(), chunk_len, sl, sr)
    starts = list(range(0, n, step))
    k = 0
    for item in it:
        length, l, r = item["stride"]
        while not (min(starts[k] + chunk_len, n) - starts[k] == length and (0 if starts[k] == 0 else sl) == l):
            k += 1
        out.append([starts[k], int(length), int(l), int(r), bool(item["is_last"])])
        k += 1
    return out


def main():
    rng = np.random.default_rng(7)
    cases = {"chunks": [], "merges": []}
    for n, c, sl, sr in [(4800000, 480000, 80000, 80000), (480000, 480000, 80000, 80000), (1, 480000, 80000, 80000),
                         (480001, 480000, 80000, 80000), (1000000, 400000, 66667, 66667), (960000, 480000, 80000, 80000),
                         (400000, 480000, 80000, 80000), (720000, 240000, 40000, 40000), (330000, 320000, 53333, 53333),
                         (1680000, 480000, 80000, 0), (1283, 100, 10, 30), (700, 100, 30, 10), (120, 100, 30, 30)]:
        cases["chunks"].append({"args": [n, c, sl, sr], "spans": spans(n, c, sl, sr)})
    for k in range(40):
        text = rng.integers(0, 50 if k % 2 else 5000, size=int(rng.integers(20, 200))).tolist()
        seqs, pos = [], 0
        while pos < len(text):
            w = int(rng.integers(8, 40))
            seq = text[max(0, pos - int(rng.integers(0, 8))): pos + w]
            if k % 3 == 0 and len(seq) > 4:      # a recognition error inside the overlap
                seq = list(seq); seq[int(rng.integers(0, 3))] = int(rng.integers(0, 5000))
            seqs.append([int(x) for x in seq])
            pos += w
        cases["merges"].append({"sequences": seqs, "merged": [int(x) for x in _find_longest_common_sequence(seqs)]})
    for seqs in ([[1, 2, 3]], [[1, 2, 3], [4, 5, 6]], [[1, 2, 3, 4], [3, 4, 5]], [[7, 7, 7, 7], [7, 7, 7]],
                 [[1, 2], [2, 3]], [[5, 6, 7, 8, 9], [6, 7, 8, 9, 10, 11], [10, 11, 12]]):
        cases["merges"].append({"sequences": seqs, "merged": [int(x) for x in _find_longest_common_sequence(seqs)]})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "longform.json")
    with open(path, "w") as f:
        json.dump(cases, f)
    print("wrote", path, len(cases["chunks"]), len(cases["merges"]))


if __name__ == "__main__":
    main()
