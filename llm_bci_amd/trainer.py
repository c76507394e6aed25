"""Native train step for the HIP NDT1: the counterpart of one iteration of the reference's
Trainer.train loop (models/trainer.py:332-362) with the same semantics —
  loss = SUM over examples, backward(loss / ga), AdamW over ALL parameters (lr, wd, eps from the
  recipe, no decay exclusions), per-step OneCycle/linear scheduler, zero_grad, DDP mean over ranks —
but without autograd, with ONE fused AdamW launch over the flat buffer, the gradient all-reduce
overlapped with backward, and loss / PER bookkeeping kept on the device (no .item() per step).

NAME2MODEL mirrors the reference registry (trainer.py:36) for the classes this package provides.
"""
import ctypes as C

import torch

from ._lib import NBCI_BF16, check, lib
from .dp import GradReducer, reduce_stats, shard_batch
from .ndt1 import NDT1, _ptr, _stream
from .schedule import LinearWarmup, OneCycle, StepDecay

from .bci import BCI  # noqa: E402
from .itransformer import iTransformer  # noqa: E402
from .patchtst import PatchTSTForSpikingActivity  # noqa: E402

NAME2MODEL = {"NDT1": NDT1, "BCI": BCI, "iTransformer": iTransformer, "PatchTST": PatchTSTForSpikingActivity}


def register_into(reference_trainer_module):
    """Registry swap: make the reference's Trainer build the HIP NDT1 for model_class 'NDT1'."""
    reference_trainer_module.NAME2MODEL["NDT1"] = NDT1
    reference_trainer_module.NAME2MODEL["BCI"] = BCI
    reference_trainer_module.NAME2MODEL["iTransformer"] = iTransformer
    reference_trainer_module.NAME2MODEL["PatchTST"] = PatchTSTForSpikingActivity


def _order(before, after):
    """`after` waits for what `before` holds now (Stream.wait_stream without the recorded event's system-scope fence: nbci_stream_order)."""
    check(lib().nbci_stream_order(C.c_void_p(before.cuda_stream), C.c_void_p(after.cuda_stream)), "nbci_stream_order")


class NativeTrainer:
    def __init__(self, model, lr=1e-3, wd=5e-5, eps=1e-8, scheduler="cosine", total_steps=1000, warmup_pct=0.0,
                 div_factor=25.0, gamma=0.95, gradient_accumulation_steps=1, betas=(0.9, 0.999), group=None,
                 compute_per=True, blank_id=0, comm_dtype="fp32", side_stream="auto", side_stream_max_rows=11500):
        self.model = model
        self.ga = gradient_accumulation_steps
        self.wd, self.eps, self.beta2 = wd, eps, betas[1]
        if scheduler == "cosine":
            self.sched = OneCycle(total_steps, lr, warmup_pct, div_factor)
        elif scheduler == "linear":
            self.sched = LinearWarmup(total_steps, lr, round(warmup_pct * total_steps), betas[0])
        elif scheduler == "step":
            self.sched = StepDecay(lr, gamma, betas[0])
        else:
            raise Exception(f"Scheduler '{scheduler}' not implemented")
        dev = model._flat.device
        n = model._total
        self.grads = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.reducer = GradReducer(model._segments, group=group, comm_dtype=comm_dtype)   # "bf16": half the bytes on the wire
        self.world = self.reducer.world
        self.group = group
        self.global_step = 1          # counts micro-batches like trainer.py:321
        self.opt_step = 0             # optimizer steps taken
        self.compute_per = compute_per
        self.blank_id = blank_id
        # device-side running stats: [loss_sum, n_examples, per_ratio_sum, n_batches]
        self.stats = torch.zeros(4, dtype=torch.float64, device=dev)
        self._per_bufs = None
        self._mstream = None   # side stream for the per-step metric (train_step)
        # One GPU: the weight-gradient GEMMs, the fold of the bias / LayerNorm gradient sums and the optimizer update (+ zero_grad) of
        # every finished segment run on a SECOND stream beside the data-gradient chain (nbci_ndt1_io.aux_stream). Same kernels, same
        # bits. Small batches leave most CUs idle inside every launch of the chain (B = 8: -7 % per step); at B = 64 it is still -2 %
        # (in-box A/B, tools/ab_side_stream.py; B = 96 = 13 728 rows: +0.8 %, every launch fills the chip and the two streams only get
        # in each other's way). "auto" = on when the model supports it and B x T' <= side_stream_max_rows (default: between the two).
        self.side_stream = side_stream
        self.side_stream_max_rows = side_stream_max_rows
        self._aux = None

    # -------------------------------------------------------------------------------------
    def _per(self, batch):
        """phoneme error rate of this batch on the device (main.py:68-74 semantics)."""
        m = self.model
        am = m.last_argmax
        B, Tp = am.shape
        tg, tl = m._io_keepalive[5], m._io_keepalive[6]
        S = tg.shape[1]
        if self._per_bufs is None or self._per_bufs[0].shape != (B, Tp) or self._per_bufs[3].numel() < B * 2 * (S + 2):
            dev = am.device
            self._per_bufs = (torch.empty(B, Tp, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev),
                              torch.empty(B, 2, dtype=torch.int32, device=dev),
                              torch.empty(B * 2 * (S + 2), dtype=torch.int32, device=dev))
        dec, dl, err, scr = self._per_bufs
        check(lib().nbci_per(_ptr(am), _ptr(tg), _ptr(tl), B, Tp, S, self.blank_id, _ptr(dec), _ptr(dl), _ptr(err), _ptr(scr),
                             _stream()), "nbci_per")
        return err

    @staticmethod
    def _per_ratio(err):
        e = err.sum(0).double()
        return e[0] / e[1]

    def _n_examples(self, loss_vec):
        """NDT1-CTC: one example per sample (ndt1.py:584); iTransformer mlm: the number of masked bins (itransformer.py:347)."""
        n = getattr(self.model, "last_n_examples", None)
        return loss_vec.numel() if n is None else n.sum().double()

    def _backward_and_reduce(self, sync):
        """Backward of the last forward into self.grads, segment by segment from the top of the flat layout down, handing every
        finished range to the reducer so its all-reduce overlaps the rest of the backward (trainer.py:339 under DDP; skipped on
        accumulation micro-steps as `no_sync` does, trainer.py:345). Models: NDT1 / iTransformer / PatchTST (segments = head,
        layers, embedder) and BCI (trainable LLM tensors, projector, then the encoder's segments)."""
        m = self.model
        nseg = len(m._segments)
        if self.reducer.world == 1 and self._use_side_stream():
            return self._backward_two_streams(sync)
        if self.reducer.world == 1 or not sync:   # nothing to overlap with: the whole backward is one call (one fold of the small-vector gradients)
            m._run_backward(self.grads, nseg - 1, 0)
            return False
        split = getattr(m, "_embed_split", None)
        if self._use_side_stream():
            # data parallel AND two streams: a segment's gradients are complete on the side stream (weight gradients, fold), so its
            # bucket goes on the wire from there; train_step then drains the buckets and updates them on the side stream as well
            main = torch.cuda.current_stream()
            if self._aux is None:
                self._aux = torch.cuda.Stream(device=self.stats.device)
            aux = self._aux
            _order(main, aux)
            for seg in range(nseg - 1, 0 if split is not None else -1, -1):
                m._run_backward(self.grads, seg, seg, aux=aux)
                with torch.cuda.stream(aux):
                    self.reducer.segment_done(self.grads, seg)
            if split is not None:
                b0, e0 = m._segments[0]
                m._run_backward(self.grads, 0, 0, embed_part=1, aux=aux)
                with torch.cuda.stream(aux):
                    self.reducer.range_done(self.grads, split, e0)
                m._run_backward(self.grads, 0, 0, embed_part=2, aux=aux)
                with torch.cuda.stream(aux):
                    self.reducer.range_done(self.grads, b0, split)
            return "aux"
        for seg in range(nseg - 1, 0 if split is not None else -1, -1):
            m._run_backward(self.grads, seg, seg)
            self.reducer.segment_done(self.grads, seg)
        if split is not None:   # the embedder in two parts: its big stack-projection bucket is on the wire during part 2
            b0, e0 = m._segments[0]
            m._run_backward(self.grads, 0, 0, embed_part=1)
            self.reducer.range_done(self.grads, split, e0)
            m._run_backward(self.grads, 0, 0, embed_part=2)
            self.reducer.range_done(self.grads, b0, split)
        return False

    def _use_side_stream(self):
        m = self.model
        if not self.side_stream or not getattr(m, "_supports_aux_stream", False) or m.compute_dtype != NBCI_BF16:
            return False
        return self.side_stream is True or getattr(m, "last_rows", 1 << 30) <= self.side_stream_max_rows

    def _adamw(self, b, e, zero=False, max_blocks=0, g_lp=None):
        """torch.optim.AdamW step (trainer.py:340) of flat range [b, e) on the current stream; zero: + zero_grad in the same pass;
        g_lp: read the gradient from this bf16 buffer (a bf16 all-reduce's staging buffer) instead of the f32 one."""
        m = self.model
        lr, beta1 = self.sched.at(self.opt_step)
        t = self.opt_step + 1
        lp = m._flat_lp if m.compute_dtype == NBCI_BF16 else None
        pw, pg, pm, pv = (x.data_ptr() for x in (m._flat, self.grads, self.m, self.v))
        plp = lp.data_ptr() if lp is not None else 0
        args = (C.c_void_p(pw + 4 * b), C.c_void_p(pg + 4 * b), C.c_void_p(pm + 4 * b), C.c_void_p(pv + 4 * b),
                C.c_void_p(plp + 2 * b) if plp else None, e - b, lr, beta1, self.beta2, self.eps, self.wd,
                1.0 - beta1 ** t, 1.0 - self.beta2 ** t, 1.0 / self.world)
        if g_lp is not None:
            args = (args[0], C.c_void_p(g_lp.data_ptr() + 2 * b)) + args[2:]
            check(lib().nbci_adamw_lp(*args, _stream()), "nbci_adamw_lp")
        elif zero:
            check(lib().nbci_adamw_zero(*args, max_blocks, _stream()), "nbci_adamw_zero")
        else:
            check(lib().nbci_adamw(*args, _stream()), "nbci_adamw")

    def _backward_two_streams(self, sync):
        """One GPU, small batch: segment by segment, the data-gradient chain on the current stream; weight gradients, the fold and -
        on a step that synchronises - AdamW + zero_grad of the finished segment on the side stream (a segment's weights are last
        read by its own backward, so its update may run while the segments below are still in their backward). Returns True
        when the optimizer step has been taken here."""
        m = self.model
        main = torch.cuda.current_stream()
        if self._aux is None:
            self._aux = torch.cuda.Stream(device=self.stats.device)
        aux = self._aux
        _order(main, aux)     # (first use; afterwards every step ends with main waiting for aux)
        for seg in range(len(m._segments) - 1, -1, -1):
            m._run_backward(self.grads, seg, seg, aux=aux)
            if sync:
                b, e = m._segments[seg]
                with torch.cuda.stream(aux):   # (256 workgroups: one per CU leaves the chain's workgroups their wave slots; A/B 256 / 512 / 1024)
                    self._adamw(b, e, zero=True, max_blocks=256)
        _order(aux, main)     # the next forward reads the updated weights and reuses the activations the weight gradients read
        return sync
    def train_step(self, batch, seed=None):
        """One micro-batch: forward, backward (+ overlapped all-reduce), and — on the steps the
        reference synchronises on (trainer.py:335) — AdamW + scheduler + zero_grad."""
        m = self.model
        m.train()
        sync = ((self.global_step - 1) % self.ga == 0)
        main = torch.cuda.current_stream()
        if self._mstream is None:
            self._mstream = torch.cuda.Stream(device=self.stats.device)
        _order(self._mstream, main)   # the previous step's metric has read its inputs before their memory is recycled
        loss_vec, preds = m._run_forward(batch, want_grad=True, seed=seed, grad_scale=1.0 / self.ga)
        # The reference computes its metric every step (trainer.py:359-362): greedy decode + PER and the loss bookkeeping read
        # forward outputs only, so they run on a second stream BESIDE the backward (two latency-bound launches, ~40 us).
        _order(main, self._mstream)
        with torch.cuda.stream(self._mstream):
            err = None
            if self.compute_per and batch.get("targets") is not None and hasattr(m, "last_argmax"):
                err = self._per(batch)
            n = getattr(m, "last_n_examples", None)
            if n is None and loss_vec.dtype == torch.float32 and loss_vec.is_contiguous():
                check(lib().nbci_step_stats(_ptr(self.stats), _ptr(loss_vec), loss_vec.numel(), float(loss_vec.numel()),
                                            _ptr(err) if err is not None else None, _stream()), "nbci_step_stats")
            else:
                self.stats[0] += loss_vec.sum().double()
                self.stats[1] += self._n_examples(loss_vec)
                if err is not None:
                    self.stats[2] += self._per_ratio(err)
                    self.stats[3] += 1
        stepped = self._backward_and_reduce(sync)
        if sync and stepped == "aux":
            # the buckets were put on the wire from the side stream: it waits for each all-reduce in turn and updates (+ clears) that
            # range, beside nothing on the main stream - which then waits for it before the next forward
            lp_grads = self.reducer.comm_bf16 and self.reducer.enabled
            done = 0
            with torch.cuda.stream(self._aux):
                for b, e in self.reducer.drain(self.grads, widen=not lp_grads):
                    if lp_grads:
                        self._adamw(b, e, g_lp=self.reducer.stage)
                    else:
                        self._adamw(b, e, zero=True)
                    done += e - b
                if lp_grads:
                    self.grads.zero_()
            if done != m._total:
                raise RuntimeError("gradient buckets do not cover the flat parameter buffer")
            _order(self._aux, main)
            self.opt_step += 1
        elif sync and stepped:
            self.opt_step += 1
        elif sync:
            # One launch per reduced bucket, in the order the buckets were put on the wire: the update of the head / layer
            # ranges runs while the embedder's all-reduce is still in flight (AdamW is elementwise: same bits as one launch).
            # drain() is lazy: the stream waits on bucket i only just before bucket i's update is queued.
            done = 0
            lp_grads = self.reducer.world > 1 and self.reducer.comm_bf16 and self.reducer.enabled and self.grads.is_cuda
            for b, e in (self.reducer.drain(self.grads, widen=not lp_grads) if self.reducer.world > 1 else [(0, m._total)]):
                self._adamw(b, e, g_lp=self.reducer.stage if lp_grads else None)   # (bf16 buckets are consumed where the all-reduce left them)
                done += e - b
            if done != m._total:
                raise RuntimeError("gradient buckets do not cover the flat parameter buffer")
            if hasattr(m, "_after_optimizer_step"):   # BCI: f32 masters of the LLM's trainable tensors -> the tensors it computes with
                m._after_optimizer_step()
            self.grads.zero_()
            self.opt_step += 1
        self.global_step += 1
        return loss_vec, preds

    @staticmethod
    def check_kernels():
        """Raises when a balanced grouped weight-gradient launch (gemm_streamk.hip) gave up waiting for a contributor: the tile it then
        stored is NaN-poisoned, so the gradients since the last check cannot be trusted. Synchronises the streams those launches ran on."""
        n = C.c_int64(0)
        check(lib().nbci_streamk_timeouts(C.byref(n)), "nbci_streamk_timeouts")
        if n.value:
            raise RuntimeError(f"{n.value} stream-K owner workgroup(s) timed out waiting for a partial tile: weight gradients were "
                               "poisoned with NaN (a stalled or starved contributor workgroup; see csrc/gemm_streamk.hip)")

    def read_stats(self, reset=True):
        """{'loss': sum_loss/sum_examples, 'PER': mean of per-batch ratios} over the steps since the
        last read (trainer.py:306-307,370-371); ONE host sync, one small all-reduce."""
        if self._mstream is not None:
            torch.cuda.current_stream().wait_stream(self._mstream)
        self.check_kernels()
        s = reduce_stats(self.stats.clone(), self.group).cpu()
        out = {"loss": (s[0] / s[1]).item() if s[1] > 0 else 0.0, "n_examples": int(s[1].item()),
               "PER": (s[2] / s[3]).item() if s[3] > 0 else None}
        if reset:
            self.stats.zero_()
        return out

    def end_epoch(self):
        if isinstance(self.sched, StepDecay):
            self.sched.end_epoch()

    # ------------------------------------------------------------------------------------- eval
    @torch.no_grad()
    def evaluate(self, batches):
        """Reference Trainer.evaluate (trainer.py:273-309): eval mode, sum-loss / sum-examples and the
        mean of per-batch PER ratios over `batches` (an iterable of device batches)."""
        m = self.model
        if self._mstream is not None:   # the last train_step's side-stream metric still reads loss / argmax / targets and _per_bufs
            torch.cuda.current_stream().wait_stream(self._mstream)
        was_training = m.training
        m.eval()
        acc = torch.zeros(4, dtype=torch.float64, device=m._flat.device)
        for batch in batches:
            loss_vec, _ = m._run_forward(batch, want_grad=False)
            acc[0] += loss_vec.sum().double()
            acc[1] += self._n_examples(loss_vec)
            if self.compute_per and batch.get("targets") is not None and hasattr(m, "last_argmax"):
                acc[2] += self._per_ratio(self._per(batch))
                acc[3] += 1
        m.train(was_training)
        s = reduce_stats(acc, self.group).cpu()
        return {"loss": (s[0] / s[1]).item() if s[1] > 0 else 0.0, "PER": (s[2] / s[3]).item() if s[3] > 0 else None}

    # ------------------------------------------------------------------------------------- resume
    def save_checkpoint(self, save_dir, rank=0):
        """Model files exactly as the reference writes them (ndt1.py:685-688) plus `trainer_state.pth`
        with what the reference never saved ("todo optimizer states", configs/trainer.yaml:11): Adam
        moments, step counters, scheduler position, RNG stream position -> exact resume. Rank 0 only."""
        import os
        if rank != 0:
            return
        if self._mstream is not None:
            torch.cuda.current_stream().wait_stream(self._mstream)
        self.check_kernels()   # never persist weights that a timed-out (NaN-poisoned) weight gradient may have reached
        self.model.save_checkpoint(save_dir)
        m = self.model
        state = {"layout": [(n, o, k) for (n, o, k, _s, _g) in m._layout], "m": self.m.cpu(), "v": self.v.cpu(),
                 "global_step": self.global_step, "opt_step": self.opt_step, "step_seed": m._step_seed,
                 "sched_epoch": getattr(self.sched, "epoch", 0), "grads": self.grads.cpu(),
                 "numerics": self._numerics()}
        if hasattr(m, "_resume_state"):   # BCI: f32 masters of the LLM's trainable (half-precision) tensors
            state["model_extra"] = m._resume_state()
        torch.save(state, os.path.join(save_dir, "trainer_state.pth"))

    def load_checkpoint(self, load_dir):
        import os
        self.model.load_checkpoint(load_dir)
        st = torch.load(os.path.join(load_dir, "trainer_state.pth"), weights_only=False)
        if [(n, o, k) for (n, o, k, _s, _g) in self.model._layout] != [tuple(x) for x in st["layout"]]:
            raise ValueError("trainer_state.pth was written for a different parameter layout")
        self.m.copy_(st["m"]); self.v.copy_(st["v"]); self.grads.copy_(st["grads"])
        self.global_step, self.opt_step = st["global_step"], st["opt_step"]
        self.model._step_seed = st["step_seed"]
        if hasattr(self.model, "_load_resume_state"):
            self.model._load_resume_state(st.get("model_extra"))
        if hasattr(self.sched, "epoch"):
            self.sched.epoch = st["sched_epoch"]
        was = st.get("numerics")
        if was is not None and was != self._numerics():
            import warnings
            warnings.warn(f"resuming with different numerics than the run that wrote {load_dir}: saved {was}, now {self._numerics()} "
                          "(compute_dtype / residual_dtype / comm_dtype change the rounding of every step from here on)")

    def _numerics(self):
        """the storage / arithmetic choices that change a run's rounding: recorded in trainer_state.pth, compared on resume"""
        names = {0: "fp32", 1: "bf16"}
        m = self.model
        return {"compute_dtype": names.get(getattr(m, "compute_dtype", 0), str(getattr(m, "compute_dtype", None))),
                "residual_dtype": names.get(getattr(m, "residual_dtype", 0), str(getattr(m, "residual_dtype", None))),
                "comm_dtype": "bf16" if self.reducer.comm_bf16 else "fp32"}
