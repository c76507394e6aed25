"""padded_array / pad_collate_fn restated (data_utils/datasets.py:191-221,236-272) — the
producer of the (B,T,N) layout the hot path consumes. Test infrastructure only."""
import numpy as np


def padded_array(arrays, dim=0, side="right", value=0, truncate=None, min_length=None):
    max_size = max(a.shape[dim] for a in arrays)
    truncate = max_size if truncate is None else truncate
    min_length = 0 if min_length is None else min_length
    assert min_length <= truncate
    pad_size = min(truncate, max(max_size, min_length))
    out = []
    for a in arrays:
        n = max(0, pad_size - a.shape[dim])
        width = [(0, 0)] * a.ndim
        width[dim] = (n, 0) if side == "left" else (0, n)
        if side not in ("left", "right"):
            raise Exception(f'"side" can only take values "right" or "left", got {side}')
        a = np.pad(a, width, mode="constant", constant_values=value)
        sl = [slice(None)] * a.ndim
        sl[dim] = slice(0, truncate)
        out.append(a[tuple(sl)])
    return np.stack(out, 0)


def make_rows(spikes_list, targets_list):
    """SpikingDatasetForDecoding.__getitem__ (datasets.py:80-97) for in-memory rows."""
    rows = []
    for s, t in zip(spikes_list, targets_list):
        rows.append({
            "spikes": s, "spikes_mask": np.ones(s.shape[0], np.int64),
            "spikes_timestamp": np.arange(0, s.shape[0]), "spikes_spacestamp": np.arange(0, s.shape[1]),
            "spikes_lengths": np.asarray(s.shape[0]), "targets": t, "targets_mask": np.ones_like(t),
            "targets_lengths": np.asarray(t.shape[0]),
        })
    return rows


CTC_PAD = {k: dict(dim=0, side="right", value=0, truncate=None, min_length=None)
           for k in ("spikes", "spikes_mask", "spikes_timestamp", "targets", "targets_mask")}


def pad_collate(rows, model_inputs, pad_dict=CTC_PAD):
    keys = rows[0].keys()
    batch, unused = {}, {}
    for k in keys:
        vals = [r[k] for r in rows]
        if isinstance(vals[0], np.ndarray):
            if k in pad_dict:
                v = padded_array(vals, **pad_dict[k]).copy()
            elif len(set(x.shape for x in vals)) == 1:
                v = np.stack(vals, 0)
            else:
                v = vals
        else:
            v = vals
        (batch if k in model_inputs else unused)[k] = v
    return batch, unused
