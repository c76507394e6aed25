"""Producer of the (B,T,N) layout the hot path consumes: the counterpart of the reference's
`padded_array` / `pad_collate_fn` (data_utils/datasets.py:191-221,236-272) and of
`SpikingDatasetForDecoding.__getitem__` (:80-97), same semantics:

  * keys listed in `pad_dict` are padded along `dim` on `side` with `value` up to the longest item
    (or `min_length`, capped by `truncate`), stacked with a leading batch axis;
  * other array keys are stacked when shapes agree, kept as lists otherwise; string arrays / non-arrays pass through;
  * the result is split into model inputs (names in the model's forward signature) and unused inputs.

Differences that matter on an MI355X node: padding writes straight into ONE pinned host buffer per key
(no per-row np.pad + np.stack + torch.clone; the buffers are recycled through a `PinnedPool`, page-locking memory per batch costs
milliseconds), `HostCollator` collates the next batches in background threads (the reference's single-process DataLoader,
trainer.py:211-222, collates on the training thread), and `DeviceFeeder` issues the H2D copies asynchronously on a side stream so
the next batch uploads while the current step runs.
"""
import collections
import concurrent.futures
import threading

import numpy as np
import torch


class PinnedPool:
    """Recycles page-locked host memory. take() hands out a (shape, dtype) VIEW of a flat pinned buffer whose capacity is the
    element count rounded up to a power of two (ragged batches change shape every step; capacity classes keep the pool small),
    give() returns it once nothing reads it any more (DeviceFeeder gives a batch's buffers back when the event behind its H2D
    copies has completed)."""

    def __init__(self):
        self._free = collections.defaultdict(list)
        self._lock = threading.Lock()
        self.allocated = 0

    def take(self, shape, dtype):
        n = 1
        for d in shape:
            n *= int(d)
        cap = 256
        while cap < n:
            cap *= 2
        with self._lock:
            flat = self._free[(cap, dtype)].pop() if self._free[(cap, dtype)] else None
            if flat is None:
                self.allocated += 1
        if flat is None:
            flat = torch.empty(cap, dtype=dtype, pin_memory=torch.cuda.is_available())
        return flat[:n].view(tuple(shape))

    def give(self, t):
        flat = t._base if t._base is not None else t
        with self._lock:
            self._free[(flat.numel(), flat.dtype)].append(flat)


def item_from_row(row, spikes_name="spikes", targets_name="targets"):
    """SpikingDatasetForDecoding.__getitem__ (datasets.py:80-97) without the deepcopy of the row."""
    out = {k: v for k, v in row.items() if k not in (spikes_name, targets_name)}
    spikes = row[spikes_name]
    out.update({
        "spikes": spikes,
        "spikes_mask": np.ones(spikes.shape[0], dtype=np.int64),
        "spikes_timestamp": np.arange(0, spikes.shape[0]),
        "spikes_spacestamp": np.arange(0, spikes.shape[1]),
        "spikes_lengths": np.asarray(spikes.shape[0]),
    })
    if targets_name in row:
        targets = row[targets_name]
        out.update({"targets": targets, "targets_mask": np.ones_like(targets), "targets_lengths": np.asarray(targets.shape[0])})
    return out


def padded_array(arrays, dim=0, side="right", value=0, truncate=None, min_length=None, pin=False, pool=None):
    """datasets.py:191-221, returning a torch tensor (optionally in pinned memory; `pool`: a PinnedPool to take the buffer from)."""
    if side not in ("left", "right"):
        raise Exception(f' "side" can only take values "right" or "left", got {side}')
    max_size = max(a.shape[dim] for a in arrays)
    truncate = max_size if truncate is None else truncate
    min_length = 0 if min_length is None else min_length
    assert min_length <= truncate, "Can't truncate below the minimum length"
    pad_size = min(truncate, max(max_size, min_length))
    shape = list(arrays[0].shape)
    shape[dim] = pad_size
    dtype = torch.from_numpy(arrays[0][:0]).dtype
    if pool is not None:
        out = pool.take([len(arrays)] + shape, dtype)
        out.fill_(value)
    elif pin and torch.cuda.is_available():
        out = torch.full([len(arrays)] + shape, value, dtype=dtype, pin_memory=True)
    else:
        out = torch.full([len(arrays)] + shape, value, dtype=dtype)
    view = out.numpy()
    for i, a in enumerate(arrays):
        n = a.shape[dim]
        # the reference pads to pad_size first and THEN keeps the first `truncate` entries along dim
        padded_len = max(n, pad_size)
        lead = padded_len - n if side == "left" else 0
        src = [slice(None)] * a.ndim
        dst = [slice(None)] * a.ndim
        lo, hi = lead, lead + n               # where the data sits in the padded row
        keep_hi = min(hi, truncate, pad_size)
        if keep_hi <= lo:
            continue
        src[dim] = slice(0, keep_hi - lo)
        dst[dim] = slice(lo, keep_hi)
        view[i][tuple(dst)] = a[tuple(src)]
    return out


def pad_collate_fn(batch, model_inputs, pad_dict, pin=False, pool=None):
    """datasets.py:236-272."""
    if isinstance(batch[0], list):
        batch = [row for sub in batch for row in sub]
    keys = batch[0].keys()
    is_arr = {k: isinstance(batch[0][k], np.ndarray) for k in keys}
    array_keys = [k for k in keys if is_arr[k] and batch[0][k].dtype.type != np.str_]
    string_keys = [k for k in keys if is_arr[k] and batch[0][k].dtype.type == np.str_]
    assert set(pad_dict.keys()).issubset(array_keys), f"Can't pad keys which are not arrays: {set(pad_dict.keys()) - set(array_keys)} "
    padded, unused = {}, {}
    for k in keys:
        vals = [row[k] for row in batch]
        if k in array_keys:
            if k in pad_dict:
                v = padded_array(vals, pin=pin, pool=pool if k in model_inputs else None, **pad_dict[k])   # (only what gets uploaded)
            elif len(set(x.shape for x in vals)) == 1:
                v = torch.from_numpy(np.stack(vals, axis=0))
            else:
                v = [torch.from_numpy(x) for x in vals]
        elif k in string_keys:
            v = np.stack(vals, axis=0)
        else:
            v = vals
        (padded if k in model_inputs else unused)[k] = v
    return padded, unused


class HostCollator:
    """Iterates `batches` (an iterable of lists of item dicts, e.g. a sampler over the dataset rows) and yields
    pad_collate_fn(batch) IN ORDER, collated up to `depth` batches ahead by `workers` background threads (the padding copies are
    numpy memcpys: they release the GIL). Padded keys land in pinned buffers from `pool`."""

    def __init__(self, batches, model_inputs, pad_dict, workers=2, depth=4, pool=None):
        self.batches = iter(batches)
        self.model_inputs, self.pad_dict = model_inputs, pad_dict
        self.pool = pool if pool is not None else PinnedPool()
        self.ex = concurrent.futures.ThreadPoolExecutor(max_workers=workers, thread_name_prefix="nbci-collate")
        self.q = collections.deque()
        self.depth = depth
        for _ in range(depth):
            self._submit()

    def _submit(self):
        try:
            rows = next(self.batches)
        except StopIteration:
            return
        self.q.append(self.ex.submit(pad_collate_fn, rows, self.model_inputs, self.pad_dict, False, self.pool))

    def __iter__(self):
        return self

    def __next__(self):
        if not self.q:
            self.ex.shutdown(wait=False)
            raise StopIteration
        out = self.q.popleft().result()
        self._submit()
        return out


class DeviceFeeder:
    """Uploads collated batches on a side stream; `next()` hands out a batch whose copies are ordered before the consumer's
    current stream. One batch is always in flight (double buffering). Pinned source buffers go back to `pool` (if given) once the
    event behind their copies has completed; the device tensors are recorded on the consumer's stream so the caching allocator
    does not recycle them while a kernel of that stream may still read them."""

    def __init__(self, iterable, device, pool=None):
        self.it = iter(iterable)
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self.pool = pool
        self._pending = collections.deque()     # (event, [pinned tensors]) whose copies may still be running
        self._next = None
        self._preload()

    def _recycle(self):
        while self._pending and self._pending[0][0].query():
            _ev, bufs = self._pending.popleft()
            for t in bufs:
                self.pool.give(t)

    def _preload(self):
        try:
            padded, unused = next(self.it)
        except StopIteration:
            self._next = None
            return
        if self.stream is None:
            self._next = (padded, unused, None)
            return
        with torch.cuda.stream(self.stream):
            dev = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in padded.items()}
            ev = torch.cuda.Event()
            ev.record(self.stream)
        if self.pool is not None:
            self._pending.append((ev, [v for v in padded.values() if torch.is_tensor(v) and v.is_pinned()]))
            self._recycle()
        self._next = (dev, unused, ev)

    def __iter__(self):
        return self

    def __next__(self):
        if self._next is None:
            raise StopIteration
        dev, unused, ev = self._next
        if ev is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            for v in dev.values():
                if torch.is_tensor(v):
                    v.record_stream(cur)
        self._preload()
        return dev, unused
