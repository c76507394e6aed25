"""Greedy CTC decode + phoneme error rate, integer-exact restatement of
utils/eval_bci.py:11-48 and the `cer` closure in main.py:68-74. Test infrastructure only.

`editdistance.eval` (third-party, unpinned, absent from /root/reference) is restated as the
textbook Levenshtein distance over token sequences.
"""
import numpy as np


def format_ctc(path, blank_id=0):
    """eval_bci.py:41-48. NOTE the reference's quirk: `last` only updates on emission, so
    A,blank,A collapses to A (standard CTC would give A,A). Returns the list of kept ids."""
    out, last = [], -1
    for idx in path:
        idx = int(idx)
        if idx != last and idx != blank_id:
            out.append(idx)
            last = idx
    return out


def edit_distance(a, b):
    """Levenshtein distance between two token sequences (editdistance.eval, eval_bci.py:14)."""
    a, b = list(a), list(b)
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i] + [0] * len(b)
        for j, y in enumerate(b, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y))
        prev = cur
    return prev[len(b)]


def per_counts(log_probs_btv, target_ids, blank_id=0):
    """main.py:68-74: argmax over ALL frames (no length masking) -> format_ctc -> errors summed
    over the batch and total target tokens. Returns (errors, n_tokens, decoded lists)."""
    paths = np.argmax(np.asarray(log_probs_btv), -1)
    errors = n = 0
    dec = []
    for pth, tgt in zip(paths, target_ids):
        d = format_ctc(pth, blank_id)
        dec.append(d)
        # word_edit_distance splits " ".join(tokens) on " ": an empty prediction becomes [""]
        # (one empty token), which editdistance counts like any other token.
        d_tok = d if len(d) else [""]
        t_tok = list(tgt) if len(tgt) else [""]
        errors += edit_distance(d_tok, t_tok)
        n += len(t_tok)
    return errors, n, dec
