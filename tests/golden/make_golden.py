"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own modules.

Runs only in the build container (needs /root/reference; never on the GPU box, never from
the test-suite). Nothing from the reference is copied: the fixtures are inputs and the
reference's outputs on them (SURVEY.md §8c, Appendix B recipe).

    python tests/golden/make_golden.py
"""
import inspect
import json
import os
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
os.chdir(REF)  # reference configs are cwd-relative (ndt1.py:17)
sys.path.insert(0, REF)

import torch  # noqa: E402
import transformers  # noqa: E402,F401
import scipy.signal  # noqa: E402
import scipy.signal.windows  # noqa: E402

scipy.signal.gaussian = scipy.signal.windows.gaussian  # ndt1.py:87 vs SciPy >= 1.13


# editdistance is not installed: exact integer Levenshtein stand-in (eval_bci.py:6,14)
class _ED:
    @staticmethod
    def eval(a, b):
        a, b = list(a), list(b)
        prev = list(range(len(b) + 1))
        for i, x in enumerate(a, 1):
            cur = [i] + [0] * len(b)
            for j, y in enumerate(b, 1):
                cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y))
            prev = cur
        return prev[-1]


sys.modules["editdistance"] = _ED

from utils.config_utils import update_config  # noqa: E402
from models.ndt1 import NDT1, create_context_mask  # noqa: E402
from data_utils.datasets import pad_collate_fn, SpikingDatasetForDecoding  # noqa: E402
from utils.eval_bci import format_ctc, word_error_count  # noqa: E402

PAD = {k: dict(dim=0, side="right", value=0, truncate=None, min_length=None)
       for k in ("spikes", "spikes_mask", "spikes_timestamp", "targets", "targets_mask")}


def build(over, seed=1):
    cfg = update_config("configs/ndt1.yaml", over)
    torch.manual_seed(seed)
    return NDT1(cfg, method_name="ctc", vocab_size=over.get("_vocab", 41), blank_id=0, zero_infinity=True)


def det(over):
    """same model config but every stochastic op off (parity is defined in this mode)."""
    o = json.loads(json.dumps(over))
    e = o.setdefault("encoder", {})
    e.setdefault("smooth_and_noise", {})["noise"] = False
    e.setdefault("embedder", {})["dropout"] = 0.0
    e.setdefault("transformer", {})["dropout"] = 0.0
    return o


def make_batch(model, lens, n_ch, tgt_lens, vocab, seed=0, days=None, blocks=None):
    g = np.random.default_rng(seed)
    rows = []
    for i, (L, S) in enumerate(zip(lens, tgt_lens)):
        rows.append({"spikes": g.standard_normal((L, n_ch)).astype(np.float32),
                     "phonemes_idx": g.integers(1, vocab, (S,)).astype(np.int64)})
        if days is not None:   # day-specific recordings (datasets.py:115-140 rows carry day_idx)
            rows[-1]["day_idx"] = np.asarray(days[i], dtype=np.int64)
        if blocks is not None:
            rows[-1]["block_idx"] = np.asarray(blocks[i], dtype=np.int64)
    ds = SpikingDatasetForDecoding(rows, targets_name="phonemes_idx")
    items = [ds[i] for i in range(len(ds))]
    names = list(inspect.signature(model.forward).parameters)
    batch, unused = pad_collate_fn(items, model_inputs=names, pad_dict=PAD)
    return rows, batch, unused


def sample_idx(n, k=64):
    return np.unique(np.linspace(0, n - 1, min(n, k)).astype(np.int64))


def summarise(t):
    a = t.detach().double().reshape(-1).numpy()
    idx = sample_idx(a.size)
    return {"sum": float(a.sum()), "abssum": float(np.abs(a).sum()), "idx": idx, "val": a[idx].astype(np.float32)}


def run_case(name, over, lens, tgt_lens, n_ch, vocab=41, full=False, steps=2, days=None, blocks=None):
    over = dict(over)
    over["_vocab"] = vocab
    ov = {k: v for k, v in over.items() if k != "_vocab"}
    model = build({**det(ov), "_vocab": vocab})
    rows, batch, unused = make_batch(model, lens, n_ch, tgt_lens, vocab, days=days, blocks=blocks)
    fx = {}
    for k, v in batch.items():
        fx["in_" + k] = v.numpy()
    inter = {}

    def hook(nm):
        def f(mod, inp, out):
            inter[nm] = out
        return f

    enc = model.encoder
    hs = [enc.smooth_and_noise.register_forward_hook(hook("smooth")),
          enc.embedder.register_forward_hook(hook("embed")),
          enc.out_norm.register_forward_hook(hook("out_norm"))]
    for i, lyr in enumerate(enc.layers):
        hs.append(lyr.register_forward_hook(hook(f"layer{i}")))
    if ov.get("encoder", {}).get("factors", {}).get("active", False):
        hs.append(enc.out_proj.register_forward_hook(hook("out_proj")))
    # --- eval forward
    model.eval()
    with torch.no_grad():
        out = model(**{k: v.clone() for k, v in batch.items()})
    fx["eval_loss"] = out.loss.numpy()
    fx["eval_preds"] = out.preds.numpy()
    fx["n_examples"] = out.n_examples.numpy()
    fx["smooth"] = inter["smooth"].numpy() if full else inter["smooth"].numpy()[:, ::7, ::5]
    fx["embed_x"] = inter["embed"][0].numpy() if full else inter["embed"][0].numpy()[:, :, ::37]
    fx["token_mask"] = inter["embed"][1].numpy()
    fx["token_ts"] = inter["embed"][2].numpy()
    fx["token_lens"] = enc.embedder.get_stacked_lens(batch["spikes_lengths"]).numpy()
    for i in range(len(enc.layers)):
        lo = inter[f"layer{i}"].numpy()
        fx[f"layer{i}_out"] = lo if full else lo[:, :, ::37]
    fx["out_norm"] = inter["out_norm"].numpy() if full else inter["out_norm"].numpy()[:, :, ::37]
    if "out_proj" in inter:   # factors projection active: the encoder output handed to the decoder (ndt1.py:372-373,450)
        fx["factors"] = inter["out_proj"].numpy() if full else inter["out_proj"].numpy()[:, :, ::37]
    # argmax path + margins + decode + PER through the reference's own metric code
    path = out.preds.argmax(-1)
    top2 = out.preds.topk(2, -1).values
    fx["argmax"] = path.numpy()
    fx["margin"] = (top2[..., 0] - top2[..., 1]).numpy()
    vocab_list = [str(i) for i in range(vocab)]
    dec = [format_ctc(p, vocab_list, 0) for p in path]
    tg = [[str(int(x)) for x in r["phonemes_idx"]] for r in rows]
    errs, n = word_error_count([" ".join(d) for d in dec], [" ".join(t) for t in tg])
    fx["per_errors"], fx["per_tokens"] = np.int64(errs), np.int64(n)
    fx["decoded_flat"] = np.array([int(x) for d in dec for x in d], np.int64)
    fx["decoded_lens"] = np.array([len(d) for d in dec], np.int64)
    for h in hs:
        h.remove()
    # --- train-mode forward/backward (stochastic ops off) + AdamW/OneCycle steps
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=5e-5, eps=1e-8)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, total_steps=100, max_lr=1e-3, pct_start=0.0, div_factor=25)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for s in range(steps):
        fx[f"lr_step{s}"] = np.float64(opt.param_groups[0]["lr"])
        fx[f"beta1_step{s}"] = np.float64(opt.param_groups[0]["betas"][0])
        out = model(**{k: v.clone() for k, v in batch.items()})
        out.loss.backward()
        if s == 0:
            fx["train_loss"] = out.loss.detach().numpy()
            for k, p in model.named_parameters():
                if full:
                    fx["grad:" + k] = p.grad.numpy().copy()
                else:
                    sm = summarise(p.grad)
                    fx["gsum:" + k] = np.array([sm["sum"], sm["abssum"]])
                    fx["gidx:" + k], fx["gval:" + k] = sm["idx"], sm["val"]
        opt.step()
        sched.step()
        opt.zero_grad()
        fx[f"loss_step{s}"] = out.loss.detach().numpy()
    for k, v in model.state_dict().items():
        if full:
            fx["w0:" + k] = sd0[k].numpy()
            fx["w2:" + k] = v.numpy()
        else:
            a, b = summarise(sd0[k]), summarise(v)
            fx["w0sum:" + k] = np.array([a["sum"], a["abssum"]])
            fx["w0idx:" + k], fx["w0val:" + k] = a["idx"], a["val"]
            fx["w2val:" + k] = b["val"]
    fx["config_json"] = np.array(json.dumps(ov))
    fx["lens"] = np.array(lens)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
    print(name, "loss", float(fx["eval_loss"]), "train_loss", float(fx["train_loss"]), "params",
          sum(p.numel() for p in model.parameters()), "token_lens", fx["token_lens"], "PER", errs, n)


def tiny(**extra):
    enc = {"embedder": {"n_channels": 16, "input_dim": 16, "max_F": 64, "stack": {"size": 4, "stride": 2}},
           "transformer": {"n_layers": 2, "hidden_size": 32, "n_heads": 2, "inter_size": 48}}
    for k, v in extra.items():
        enc.setdefault(k, {}).update(v)
    return {"encoder": enc}


def ctc_cases():
    torch.manual_seed(0)
    fx = {}
    cases = [  # (T, V, targets, in_len, tgt_len)
        ("basic", 12, 6, [[1, 2, 3], [2, 2, 4]], [12, 10], [3, 3]),
        ("repeat_infeasible", 5, 5, [[1, 1, 1], [1, 2, 0]], [5, 5], [3, 2]),       # 1,1,1 needs 5 frames: feasible edge
        ("too_short", 4, 5, [[1, 1, 1], [3, 0, 0]], [4, 2], [3, 1]),                # first infeasible -> zero_infinity
        ("empty_target", 6, 4, [[0, 0], [2, 0]], [6, 3], [0, 1]),
        ("long", 40, 41, [list(range(1, 16)), [5] * 15], [40, 33], [15, 15]),
    ]
    for nm, T, V, tg, il, tl in cases:
        lp = torch.randn(len(tg), T, V).log_softmax(-1).requires_grad_(True)
        loss = torch.nn.CTCLoss(reduction="none", blank=0, zero_infinity=True)(
            lp.transpose(0, 1), torch.tensor(tg), torch.tensor(il), torch.tensor(tl))
        loss.sum().backward()
        fx[nm + "_lp"] = lp.detach().numpy(); fx[nm + "_targets"] = np.array(tg); fx[nm + "_il"] = np.array(il)
        fx[nm + "_tl"] = np.array(tl); fx[nm + "_loss"] = loss.detach().numpy(); fx[nm + "_grad"] = lp.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "ctc_cases.npz"), **fx)
    print("ctc_cases", {k: v.tolist() for k, v in fx.items() if k.endswith("_loss")})


def metric_cases():
    """format_ctc / word_error_count on hand-made paths (A blank A, repeated blanks, pad frames)."""
    vocab = [str(i) for i in range(41)]
    paths = [[3, 0, 3, 3, 0, 0, 5, 5, 0, 3], [0, 0, 0, 0], [7, 7, 7, 7], [1, 2, 1, 2, 0, 2, 0, 1], [4, 0, 0, 0, 0, 0, 0, 0]]
    tgts = [[3, 3, 5, 3], [1, 2], [7], [1, 2, 1, 2, 2, 1], [4, 9]]
    fx = {"paths_flat": np.array([x for p in paths for x in p]), "paths_len": np.array([len(p) for p in paths]),
          "tgts_flat": np.array([x for t in tgts for x in t]), "tgts_len": np.array([len(t) for t in tgts])}
    dec = [format_ctc(torch.tensor(p), vocab, 0) for p in paths]
    fx["dec_flat"] = np.array([int(x) for d in dec for x in d]); fx["dec_len"] = np.array([len(d) for d in dec])
    per = []
    for d, t in zip(dec, tgts):
        e, n = word_error_count(" ".join(d), " ".join(str(x) for x in t))
        per.append((e, n))
    fx["per"] = np.array(per)
    np.savez_compressed(os.path.join(OUT, "metric_cases.npz"), **fx)
    print("metric_cases", per)


def misc_cases():
    fx = {}
    for f, b in ((-2, -2), (-1, -1), (0, -2), (3, 2), (-2, 0), (0, 0), (5, -1)):
        fx[f"ctx_{f}_{b}"] = create_context_mask(f, b, 24).numpy()
    np.savez_compressed(os.path.join(OUT, "misc_cases.npz"), **fx)


if __name__ == "__main__" and "--tokens" in sys.argv:
    # learned prefix tokens (ndt1.py:151-155,192-201,444-448): [day, block, spike tokens...], stripped after out_norm
    run_case("g_tiny_tokens", tiny(embedder={"day_token": True, "block_token": True, "n_days": 3, "n_blocks": 4}), [30, 22, 17, 26],
             [5, 4, 2, 3], 16, vocab=11, full=True, days=[2, 0, 2, 1], blocks=[3, 3, 0, 1])
    run_case("g_tiny_daytoken", tiny(embedder={"day_token": True, "n_days": 3}, context={"forward": 3, "backward": 2}), [30, 22, 17],
             [5, 4, 2], 16, vocab=11, full=True, days=[1, 1, 0])
    sys.exit(0)

if __name__ == "__main__" and "--adapt" in sys.argv:
    # embedder.adapt: one embed_spikes Linear per recording day, picked per sample by day_idx (ndt1.py:124-129,170-171)
    run_case("g_tiny_adapt", tiny(embedder={"adapt": True, "n_days": 3}), [30, 22, 17, 26], [5, 4, 2, 3], 16, vocab=11, full=True,
             days=[2, 0, 2, 1])
    sys.exit(0)

if __name__ == "__main__" and "--factors" in sys.argv:
    # NeuralFactorsProjection active (ndt1.py:348-373): Linear(hidden -> size) + act between out_norm and the decoder
    run_case("g_tiny_factors", tiny(factors={"active": True, "size": 24, "act": "relu", "bias": True}), [30, 22, 17], [5, 4, 2], 16,
             vocab=11, full=True)
    run_case("g_tiny_factors_fix", tiny(factors={"active": True, "size": 40, "act": "tanh", "bias": True, "fixup_init": True,
                                                 "init_range": 0.1}), [30, 22, 17], [5, 4, 2], 16, vocab=11, full=True)
    sys.exit(0)

if __name__ == "__main__" and "--rope" in sys.argv:
    # RoPE (ndt1.py:46-71,285-286) at real widths: C1 (2 layers x 1024, head 128, ragged, T' = 18 / 10 - the one-workgroup attention)
    # and a long case (T' = 293 / 218 - the streaming attention), positions = the stacked timestamps
    rope = {"encoder": {"embedder": {"n_channels": 64}, "transformer": {"n_layers": 2, "use_rope": True}}}
    run_case("g_c1_rope", rope, [100, 70], [10, 6], 64)
    run_case("g_long_rope", rope, [1200, 900], [80, 60], 64)
    sys.exit(0)

if __name__ == "__main__" and "--long" in sys.argv:
    # sequences beyond the one-workgroup attention kernel (T' > 160): 1200 bins -> 293 tokens, ragged, 2 layers x 1024 (head 128)
    c1 = {"encoder": {"embedder": {"n_channels": 64}, "transformer": {"n_layers": 2}}}
    run_case("g_long", c1, [1200, 900], [80, 60], 64)
    c1c = json.loads(json.dumps(c1)); c1c["encoder"]["context"] = {"forward": 5, "backward": 40}
    run_case("g_long_ctx", c1c, [1200, 900], [80, 60], 64)

if __name__ == "__main__" and not any(f in sys.argv for f in ("--bci", "--itr", "--ptst", "--masker-copy", "--long", "--itr-region", "--itr-uni",
                                                             "--itr-wide", "--ckpt", "--ckpt-itr", "--rope")):
    run_case("g_tiny", tiny(), [30, 22, 17], [5, 4, 2], 16, vocab=11, full=True)
    run_case("g_tiny_ctx", tiny(context={"forward": 3, "backward": 2}), [30, 22, 17], [5, 4, 2], 16, vocab=11, full=True)
    run_case("g_tiny_rope", tiny(transformer={"use_rope": True}), [30, 22, 17], [5, 4, 2], 16, vocab=11, full=True)
    run_case("g_c1", {"encoder": {"embedder": {"n_channels": 64}, "transformer": {"n_layers": 2}}}, [100, 70], [10, 6], 64)
    run_case("g_c2", {}, [600, 450], [60, 40], 256)
    ctc_cases()
    metric_cases()
    misc_cases()


def bci_case():
    """BCI.prepare_embeds (models/bci.py:107-168) with the debug tiny Llama: embeddings / mask / targets after the
    splice, and gradients of sum(input_embeds * R) wrt projector + encoder parameters."""
    import types
    peft = types.ModuleType("peft"); peft.LoraConfig = None; peft.get_peft_model = None
    sys.modules["peft"] = peft
    from models.bci import BCI
    enc = tiny()["encoder"]
    enc = json.loads(json.dumps(enc))
    enc.setdefault("smooth_and_noise", {})["noise"] = False
    enc["embedder"]["dropout"] = 0.0; enc["transformer"]["dropout"] = 0.0
    cfg = {"projector": {"stacking": 2, "inter_size": 48, "bias": True, "act": "relu"}, "ndt1": {"encoder": enc}}
    torch.manual_seed(3)
    m = BCI(cfg, llm_path=None, debug=True, method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True)
    m.llm.float()   # keep the text embeddings exact for the fixture (the reference runs them in fp16)
    m.eval()
    g = np.random.default_rng(4)
    B, T, Lt = 3, 32, 6     # T' = 15 -> padded to 16 for stacking 2
    lens = [32, 26, 19]
    spikes = g.standard_normal((B, T, 16)).astype(np.float32)
    smask = np.zeros((B, T), np.int64); ts = np.zeros((B, T), np.int64)
    for b, L in enumerate(lens):
        spikes[b, L:] = 0; smask[b, :L] = 1; ts[b, :L] = np.arange(L)
    input_ids = g.integers(0, 100, (B, Lt)).astype(np.int64)
    amask = np.ones((B, Lt), np.int64); amask[2, 4:] = 0
    split = np.array([1, 3, 0], np.int64)
    targets = g.integers(0, 100, (B, Lt)).astype(np.int64); targets[:, :2] = -100
    tt = lambda a: torch.from_numpy(a)
    emb, mask, tg = m.prepare_embeds(tt(input_ids), tt(amask), tt(split), tt(spikes), tt(smask), tt(ts), tt(np.array(lens)), None, None,
                                     tt(targets))
    R = torch.from_numpy(g.standard_normal(tuple(emb.shape)).astype(np.float32))
    (emb * R).sum().backward()
    fx = {"spikes": spikes, "spikes_mask": smask, "spikes_timestamp": ts, "spikes_lengths": np.array(lens), "input_ids": input_ids,
          "attention_mask": amask, "input_split": split, "targets": targets, "R": R.numpy(),
          "out_embeds": emb.detach().numpy(), "out_mask": mask.numpy(), "out_targets": tg.numpy(),
          "embed_table": m.llm.get_input_embeddings().weight.detach().numpy()[:100].copy(),
          "config_json": np.array(json.dumps(cfg))}
    for k, v in m.ndt1.state_dict().items():
        fx["w:ndt1." + k] = v.numpy()
    for k, v in m.projector.state_dict().items():
        fx["w:projector." + k] = v.numpy()
    for k, p in m.projector.named_parameters():
        fx["g:projector." + k] = p.grad.numpy()
    for k, p in m.ndt1.named_parameters():
        if p.grad is not None and any(s in k for s in ("out_norm", "layers.1.mlp.down_proj", "embed_spikes", "layers.0.attn.query")):
            fx["g:ndt1." + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "g_bci.npz"), **fx)
    print("g_bci", emb.shape, mask.tolist(), tg.tolist()[0])


def bci_forward_case():
    """BCI.forward (models/bci.py:173-219) end to end through the reference's own `llm=` constructor argument (bci.py:48-49) with
    a 2-layer random Llama small enough to store whole: loss (shifted CE sum), n_examples, logits, and d loss / d parameters of
    the projector, the encoder and the LLM; once with the LLM in fp32 (tight parity of the coupler) and once exactly as the
    reference runs it (llm.to(float16), bci.py:71,190)."""
    import types
    peft = types.ModuleType("peft"); peft.LoraConfig = None; peft.get_peft_model = None
    sys.modules["peft"] = peft
    from transformers import AutoModelForCausalLM, LlamaConfig
    from models.bci import BCI
    enc = json.loads(json.dumps(tiny()["encoder"]))
    enc.setdefault("smooth_and_noise", {})["noise"] = False
    enc["embedder"]["dropout"] = 0.0; enc["transformer"]["dropout"] = 0.0
    cfg = {"projector": {"stacking": 2, "inter_size": 48, "bias": True, "act": "relu"}, "ndt1": {"encoder": enc}}
    llm_cfg = dict(vocab_size=128, hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4,
                   num_key_value_heads=4, max_position_embeddings=64)
    torch.manual_seed(5)
    llm = AutoModelForCausalLM.from_config(LlamaConfig(**llm_cfg))
    m = BCI(cfg, llm_path=None, llm=llm, method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True)
    g = np.random.default_rng(6)
    B, T, Lt = 3, 32, 7
    lens = [32, 27, 18]
    spikes = g.standard_normal((B, T, 16)).astype(np.float32)
    smask = np.zeros((B, T), np.int64); ts = np.zeros((B, T), np.int64)
    for b, L in enumerate(lens):
        spikes[b, L:] = 0; smask[b, :L] = 1; ts[b, :L] = np.arange(L)
    input_ids = g.integers(0, 128, (B, Lt)).astype(np.int64)
    amask = np.ones((B, Lt), np.int64); amask[1, 5:] = 0
    split = np.array([2, 3, 1], np.int64)
    targets = g.integers(0, 128, (B, Lt)).astype(np.int64); targets[:, :3] = -100; targets[1, 5:] = -100
    tt = lambda a: torch.from_numpy(a)
    args = (tt(input_ids), tt(amask), tt(split), tt(spikes), tt(smask), tt(ts), tt(np.array(lens)), None, None, tt(targets))
    fx = {"spikes": spikes, "spikes_mask": smask, "spikes_timestamp": ts, "spikes_lengths": np.array(lens), "input_ids": input_ids,
          "attention_mask": amask, "input_split": split, "targets": targets, "config_json": np.array(json.dumps(cfg)),
          "llm_config_json": np.array(json.dumps(llm_cfg))}
    m.eval()
    m.llm.float()
    for k, v in m.ndt1.state_dict().items():
        fx["w:ndt1." + k] = v.numpy().copy()
    for k, v in m.projector.state_dict().items():
        fx["w:projector." + k] = v.numpy().copy()
    for k, v in m.llm.state_dict().items():
        fx["w:llm." + k] = v.numpy().copy()
    out = m(*args)
    out.loss.backward()
    fx.update(f32_loss=np.float64(out.loss.item()), n_examples=np.int64(out.n_examples.item()), f32_logits=out.preds.detach().numpy(),
              out_targets=out.targets.numpy())
    for k, p in m.projector.named_parameters():
        fx["g32:projector." + k] = p.grad.numpy().copy()
    for k, p in m.ndt1.named_parameters():
        if p.grad is not None and any(s in k for s in ("out_norm", "layers.1.mlp.down_proj", "embed_spikes", "layers.0.attn.query", "stack_projection.bias")):
            fx["g32:ndt1." + k] = p.grad.numpy().copy()
    for k, p in m.llm.named_parameters():
        if any(s in k for s in ("embed_tokens", "layers.1.self_attn.q_proj", "layers.0.mlp.down_proj", "lm_head", "norm.weight")):
            fx["g32:llm." + k] = p.grad.numpy().copy()
    m.zero_grad()
    m.llm.to(torch.float16)          # the reference's own precision for the LLM (bci.py:71)
    out16 = m(*args)
    out16.loss.backward()
    fx.update(f16_loss=np.float64(out16.loss.item()), f16_logits=out16.preds.detach().float().numpy())
    for k, p in m.projector.named_parameters():
        fx["g16:projector." + k] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g_bci_fwd.npz"), **fx)
    print("g_bci_fwd", out.loss.item(), out16.loss.item(), int(out.n_examples), tuple(out.preds.shape))


if __name__ == "__main__" and "--bci" in sys.argv:
    bci_case()
    bci_forward_case()


# ------------------------------------------------------------------------------------------------
# iTransformer SSL (models/itransformer.py mlm branch, masker.py) — `--itr`
# ------------------------------------------------------------------------------------------------
def _install_torchvision_mlp():
    """torchvision is not installed here. torchvision.ops.MLP (third party, version unpinned by the
    reference) is, by its published definition, Sequential[Linear, (norm), act, Dropout]* + Linear + Dropout."""
    import types

    class MLP(torch.nn.Sequential):
        def __init__(self, in_channels, hidden_channels, norm_layer=None, activation_layer=torch.nn.ReLU, inplace=None,
                     bias=True, dropout=0.0):
            params = {} if inplace is None else {"inplace": inplace}
            layers, d = [], in_channels
            for h in hidden_channels[:-1]:
                layers.append(torch.nn.Linear(d, h, bias=bias))
                if norm_layer is not None:
                    layers.append(norm_layer(h))
                layers.append(activation_layer(**params))
                layers.append(torch.nn.Dropout(dropout, **params))
                d = h
            layers.append(torch.nn.Linear(d, hidden_channels[-1], bias=bias))
            layers.append(torch.nn.Dropout(dropout, **params))
            super().__init__(*layers)

    tv = types.ModuleType("torchvision"); ops = types.ModuleType("torchvision.ops")
    ops.MLP = MLP; tv.ops = ops
    sys.modules["torchvision"] = tv; sys.modules["torchvision.ops"] = ops


def itr_case(name, over, B, N, lens, full, steps=2, log_input=True, loss="poisson_nll", spacestamp=False, regions=False, depths=False):
    """regions: per-neuron brain-region names drawn from over["encoder"]["regions"] (embed_region, itransformer.py:133-141,195-198);
    depths: per-neuron depths (embed_depth, :143-150,200-202). Both are forward() inputs, recorded in the fixture."""
    from models.itransformer import iTransformer
    cfg = update_config("configs/itransformer.yaml", over)
    torch.manual_seed(1)
    model = iTransformer(cfg, method_name="mlm", log_input=log_input, loss=loss)
    T = model.config.encoder.embedder.max_n_bins
    g = np.random.default_rng(0)
    spikes = g.poisson(0.5, (B, T, N)).astype(np.float32)
    smask = np.zeros((B, T), np.int64); ts = np.zeros((B, T), np.int64)
    for b, L in enumerate(lens):          # left padding (trainer_ssl_itransformer.yaml:66-90)
        spikes[b, :T - L] = 0; smask[b, T - L:] = 1; ts[b, T - L:] = np.arange(L)
    batch = {"spikes": torch.from_numpy(spikes), "spikes_mask": torch.from_numpy(smask), "spikes_timestamp": torch.from_numpy(ts)}
    if spacestamp:
        ss = np.stack([g.permutation(model.config.encoder.max_n_channels)[:N] for _ in range(B)]).astype(np.int64)
        batch["spikes_spacestamp"] = torch.from_numpy(ss)
    fx = {"in_" + k: v.numpy() for k, v in batch.items()}
    if regions:
        names = list(over["encoder"]["regions"])
        nr = np.array(names)[g.integers(0, len(names), (B, N))]
        batch["neuron_regions"] = nr                                   # np.ndarray of str, as datasets.py hands it over
        fx["in_neuron_regions"] = nr.astype("U16")
    if depths:
        nd = (g.uniform(0.0, 3.84, (B, N))).astype(np.float32)          # probe depth in mm
        batch["neuron_depths"] = torch.from_numpy(nd)
        fx["in_neuron_depths"] = nd
    inter, masks = {}, []

    def hook(nm):
        def f(mod, inp, out):
            inter[nm] = out
        return f

    enc = model.encoder
    uni = enc.mode == "transformer"
    # (mlp mode: the hooked output of `embed` is the tensor the in-place `tokens += ...` adds then land in; transformer mode: `embed` =
    #  the UnivariateTransformer's CLS outputs (B,N,h), `embed_proj` = the tensor the adds land in)
    hs = [enc.embed.register_forward_hook(hook("emb_cls" if uni else "embed")), enc.embed_dropout.register_forward_hook(hook("tokens")),
          enc.transformer.register_forward_hook(hook("encoder"))]
    if uni:
        hs.append(enc.embed_proj.register_forward_hook(hook("embed")))
        hs.append(enc.embed.transformer.register_forward_hook(hook("emb_out")))
        for i, lyr in enumerate(enc.embed.transformer.layers):
            hs.append(lyr.register_forward_hook(hook(f"emb_layer{i}")))
    for i, lyr in enumerate(enc.transformer.layers):
        hs.append(lyr.register_forward_hook(hook(f"layer{i}")))
    for mk in model.masker.values():
        hs.append(mk.register_forward_hook(lambda mod, inp, out: masks.append(out[1].numpy().copy())))
    model.eval()
    with torch.no_grad():
        out = model(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
    cut = (lambda a: a) if full else (lambda a: a[..., ::37])
    fx["eval_raw_mask"] = masks[-1]
    fx["eval_loss"] = out.loss.numpy(); fx["eval_n_examples"] = out.n_examples.numpy()
    fx["eval_mask"] = out.mask.numpy()
    fx["eval_preds"] = out.preds.numpy() if full else out.preds.numpy()[:, ::3, ::5]
    fx["eval_targets_sum"] = np.float64(out.targets.double().sum().item())
    fx["embed"] = cut(inter["embed"].numpy()); fx["tokens"] = cut(inter["tokens"].numpy())
    for i in range(len(enc.transformer.layers)):
        fx[f"layer{i}_out"] = cut(inter[f"layer{i}"].numpy())
    fx["encoder_out"] = cut(inter["encoder"].numpy())
    if uni:
        ecut = (lambda a: a) if full else (lambda a: a[::5, ::7, ::9])      # (B*N, T+1, h)
        fx["emb_cls"] = inter["emb_cls"].numpy(); fx["emb_out"] = ecut(inter["emb_out"].numpy())
        for i in range(len(enc.embed.transformer.layers)):
            fx[f"emb_layer{i}_out"] = ecut(inter[f"emb_layer{i}"].numpy())
    for h in hs[:-len(model.masker)]:
        h.remove()
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.01, eps=1e-8)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, total_steps=100, max_lr=1e-4, pct_start=0.15, div_factor=25)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for s in range(steps):
        fx[f"lr_step{s}"] = np.float64(opt.param_groups[0]["lr"]); fx[f"beta1_step{s}"] = np.float64(opt.param_groups[0]["betas"][0])
        out = model(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
        fx[f"raw_mask_step{s}"] = masks[-1]
        out.loss.backward()
        fx[f"loss_step{s}"] = out.loss.detach().numpy(); fx[f"n_examples_step{s}"] = out.n_examples.numpy()
        if s == 0:
            for k, p in model.named_parameters():
                if full:
                    fx["grad:" + k] = p.grad.numpy().copy()
                else:
                    sm = summarise(p.grad)
                    fx["gsum:" + k] = np.array([sm["sum"], sm["abssum"]]); fx["gidx:" + k], fx["gval:" + k] = sm["idx"], sm["val"]
        opt.step(); sched.step(); opt.zero_grad()
    for k, v in model.state_dict().items():
        if full:
            fx["w0:" + k] = sd0[k].numpy(); fx["w2:" + k] = v.numpy()
        else:
            a, b = summarise(sd0[k]), summarise(v)
            fx["w0sum:" + k] = np.array([a["sum"], a["abssum"]]); fx["w0idx:" + k], fx["w0val:" + k] = a["idx"], a["val"]
            fx["w2val:" + k] = b["val"]
    fx["config_json"] = np.array(json.dumps(over)); fx["lens"] = np.array(lens)
    fx["kwargs_json"] = np.array(json.dumps({"log_input": log_input, "loss": loss}))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
    print(name, "eval loss", float(fx["eval_loss"]), "n", int(fx["eval_n_examples"]), "train loss", float(fx["loss_step0"]),
          "params", sum(p.numel() for p in model.parameters()), [k for k in sd0][:6])


def masker_cases():
    """Masker.forward (models/masker.py:44-104): shapes / value rules per mode, recorded from the reference with a fixed torch seed.
    (The draws themselves are torch's Philox/MT stream and are not reproducible elsewhere; the fixture pins the STRUCTURE:
    which axes a mask is constant along, zero/random replacement rules, eval/inactive pass-through, expand_timesteps.)"""
    from models.masker import Masker
    from utils.config_utils import DictConfig
    fx = {}
    g = np.random.default_rng(5)
    spikes = g.poisson(2.0, (4, 20, 12)).astype(np.float32)
    base = dict(force_active=True, active=True, ratio=0.3, zero_ratio=1.0, random_ratio=1.0, expand_prob=0.0, max_timespan=1,
                regions=None, channels=None)
    for mode, extra in (("temporal", {}), ("neuron", {}), ("random", {}), ("co-smooth", {"channels": [1, 5, 7]}),
                        ("temporal_expand", {"mode": "temporal", "expand_prob": 1.0, "max_timespan": 3}),
                        ("random_mixed", {"mode": "random", "zero_ratio": 0.5, "random_ratio": 0.5})):
        c = dict(base); c["mode"] = mode; c.update(extra)
        torch.manual_seed(7)
        mk = Masker(DictConfig(c)); mk.train()
        out, mask = mk(torch.from_numpy(spikes.copy()))
        fx[mode + "_out"] = out.numpy(); fx[mode + "_mask"] = mask.numpy(); fx[mode + "_cfg"] = np.array(json.dumps(c))
    w = torch.zeros(2, 11); w[0, 3] = 1; w[1, 0] = 1; w[1, 10] = 1
    for width in (1, 2, 3, 4):
        fx[f"expand_{width}"] = Masker.expand_timesteps(w, width).numpy()
    fx["expand_in"] = w.numpy(); fx["spikes"] = spikes
    np.savez_compressed(os.path.join(OUT, "masker_cases.npz"), **fx)
    print("masker_cases", {k: v.shape for k, v in fx.items() if k.endswith("_mask")})


def masker_copy_cases():
    """The three extra modes of the reference's "models/masker copy.py" (forward-pred :81-85, inter-region :86-94, intra-region
    :95-104 + :133). The file name has a space, so it is loaded by path; Python's `random` (its region sample) and torch are seeded.
    As for masker_cases the draws are not reproducible elsewhere: the fixture pins the STRUCTURE (which bins are masked / returned
    as targets / corrupted) in cases where it is deterministic (ratio 1, every listed region sampled) and the invariants elsewhere."""
    import importlib.util
    import random
    spec = importlib.util.spec_from_file_location("masker_copy", os.path.join(REF, "models", "masker copy.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from utils.config_utils import DictConfig
    fx = {}
    g = np.random.default_rng(6)
    B, T, N = 4, 10, 12
    spikes = g.poisson(2.0, (B, T, N)).astype(np.float32) + 1.0      # strictly positive: a zeroed bin is recognisable
    regions = np.array([["CA1", "DG", "PO", "VIS"][(i * 7 + i // 5) % 4] for i in range(B * N)]).reshape(B, N)
    base = dict(force_active=True, mode="neuron", ratio=1.0, zero_ratio=1.0, random_ratio=1.0, expand_prob=0.0, max_timespan=1,
                channels=None, timesteps=None, mask_regions=None, target_regions=None, n_mask_regions=1)
    cases = {"forward_pred": dict(mode="forward-pred", timesteps=[2, 5, 6]),
             "inter_all": dict(mode="inter-region", mask_regions=["CA1", "PO"], n_mask_regions=2, ratio=1.0),
             "inter_half": dict(mode="inter-region", mask_regions=["CA1", "PO", "DG"], n_mask_regions=1, ratio=0.5),
             "intra_all": dict(mode="intra-region", target_regions=["DG"], n_mask_regions=1, ratio=1.0),
             "intra_some": dict(mode="intra-region", target_regions=["DG", "VIS"], n_mask_regions=2, ratio=0.4)}
    for name, extra in cases.items():
        c = dict(base); c.update(extra)
        torch.manual_seed(7); random.seed(7)
        mk = mod.Masker(DictConfig(c)); mk.train()
        out, mask = mk(torch.from_numpy(spikes.copy()), regions)
        fx[name + "_out"] = out.numpy(); fx[name + "_mask"] = mask.numpy(); fx[name + "_cfg"] = np.array(json.dumps(c))
    fx["spikes"] = spikes; fx["regions"] = regions.astype("U8")
    np.savez_compressed(os.path.join(OUT, "masker_copy_cases.npz"), **fx)
    print("masker_copy_cases", {k: int(v.sum()) for k, v in fx.items() if k.endswith("_mask")})


def itr_tiny(**enc_extra):
    enc = {"embedder": {"max_n_bins": 12, "dropout": 0.0}, "hidden_size": 32, "n_heads": 2, "n_layers": 2, "dropout": 0.0,
           "max_n_channels": 16, "embed_region": False}
    enc.update(enc_extra)
    return {"encoder": enc, "masker": {"main": {"active": True, "regions": None, "ratio": 0.3}}}


if __name__ == "__main__" and "--masker-copy" in sys.argv:
    masker_copy_cases()

if __name__ == "__main__" and "--itr" in sys.argv:
    _install_torchvision_mlp()
    itr_case("g_itr_tiny", itr_tiny(), 3, 10, [12, 9, 7], full=True)
    itr_case("g_itr_tiny_ss", itr_tiny(), 3, 10, [12, 12, 5], full=True, spacestamp=True)
    itr_case("g_itr_tiny_rate", itr_tiny(), 3, 10, [12, 9, 7], full=True, log_input=False)
    itr_case("g_itr_tiny_mse", itr_tiny(), 3, 10, [12, 9, 7], full=True, loss="mse")
    itr_case("g_itr_c3", {"encoder": {"embedder": {"dropout": 0.0}, "dropout": 0.0, "embed_region": False},
                          "masker": {"main": {"active": True, "regions": None}}}, 4, 64, [100, 100, 80, 61], full=False)
    masker_cases()


if __name__ == "__main__" and "--itr-region" in sys.argv:
    # embed_region: true - the shipped default of configs/itransformer.yaml:38 (main.py:39-42 fills `regions` from the data) - and embed_depth
    _install_torchvision_mlp()
    REGS = ["CA1", "DG", "LP", "PO", "VISa"]
    itr_case("g_itr_tiny_region", itr_tiny(embed_region=True, regions=REGS), 3, 10, [12, 9, 7], full=True, regions=True)
    itr_case("g_itr_tiny_region_depth", itr_tiny(embed_region=True, regions=REGS[:3], embed_depth=True), 3, 10, [12, 12, 5], full=True,
             regions=True, depths=True, spacestamp=True)
    itr_case("g_itr_tiny_depth", itr_tiny(embed_depth=True), 3, 10, [12, 9, 7], full=True, depths=True)
    itr_case("g_itr_c3w_region", {"encoder": {"embedder": {"dropout": 0.0}, "dropout": 0.0, "embed_region": True, "regions": REGS + ["VISp", "TH", "ZI"]},
                                  "masker": {"main": {"active": True, "regions": None}}}, 2, 668, [100, 73], full=False, regions=True)

if __name__ == "__main__" and "--itr-uni" in sys.argv:
    # embedder.mode: transformer - the UnivariateTransformer embedder (itransformer.py:40-93) + embed_proj (:119-124)
    _install_torchvision_mlp()
    emb = {"mode": "transformer", "max_n_bins": 12, "dropout": 0.0, "hidden_size": 16, "n_heads": 2, "n_layers": 2, "activation": "relu"}
    if "--only-c3" not in sys.argv:
        itr_case("g_itr_tiny_uni", itr_tiny(embedder=emb), 3, 10, [12, 9, 7], full=True)
    if "--only-c3" not in sys.argv:
        itr_case("g_itr_tiny_uni_all", itr_tiny(embedder=emb, embed_region=True, regions=["CA1", "DG", "LP"], embed_depth=True), 3, 10, [12, 12, 5],
                 full=True, regions=True, depths=True, spacestamp=True, loss="mse")
    # the shipped embedder widths (configs/itransformer.yaml:21-27: 128 x 4 heads x 4 layers, 100 bins) under the shipped encoder, few channels
    itr_case("g_itr_uni_c3", {"encoder": {"embedder": {"mode": "transformer", "dropout": 0.0}, "dropout": 0.0, "embed_region": False},
                              "masker": {"main": {"active": True, "regions": None}}}, 2, 24, [100, 61], full=False)

if __name__ == "__main__" and "--itr-wide" in sys.argv:   # the recipe's channel count (configs/itransformer.yaml: 668 channels): sampled fixture
    _install_torchvision_mlp()
    itr_case("g_itr_c3w", {"encoder": {"embedder": {"dropout": 0.0}, "dropout": 0.0, "embed_region": False},
                           "masker": {"main": {"active": True, "regions": None}}}, 2, 668, [100, 73], full=False)


# ------------------------------------------------------------------------------------------------
# PatchTST (models/patchtst.py over transformers' PatchTSTModel) — `--ptst`
# ------------------------------------------------------------------------------------------------
def ptst_case(name, over, method, B, lens, tgt_lens=None, vocab=11, full=True, steps=2, log_input=True, loss="poisson_nll"):
    from models.patchtst import PatchTSTForSpikingActivity
    cfg = update_config("configs/patchtst.yaml", over)
    torch.manual_seed(1)
    kw = dict(method_name=method)
    if method == "ctc":
        kw.update(vocab_size=vocab, blank_id=0, zero_infinity=True)
    else:
        kw.update(log_input=log_input, loss=loss)
    model = PatchTSTForSpikingActivity(cfg, **kw)
    ec = model.config.encoder
    T, C = ec["context_length"], ec["num_input_channels"]
    g = np.random.default_rng(0)
    spikes = (g.poisson(0.8, (B, T, C)) if method == "mlm" else g.standard_normal((B, T, C))).astype(np.float32)
    smask = np.zeros((B, T), np.int64)
    for b, L in enumerate(lens):          # right padding (trainer_ctc_ndt1.yaml-style collate)
        spikes[b, L:] = 0; smask[b, :L] = 1
    batch = {"spikes": torch.from_numpy(spikes), "spikes_mask": torch.from_numpy(smask), "spikes_lengths": torch.tensor(lens)}
    if method == "ctc":
        S = max(tgt_lens)
        tg = np.zeros((B, S), np.int64)
        for b, n in enumerate(tgt_lens):
            tg[b, :n] = g.integers(1, vocab, n)
        batch["targets"] = torch.from_numpy(tg); batch["targets_lengths"] = torch.tensor(tgt_lens)
    fx = {"in_" + k: v.numpy() for k, v in batch.items()}
    inter, masks = {}, []
    enc = model.encoder
    hs = [enc.encoder.positional_encoder.register_forward_hook(lambda m, i, o: inter.__setitem__("embed", o.detach()))]
    for i, lyr in enumerate(enc.encoder.layers):
        hs.append(lyr.register_forward_hook(lambda m, inp, o, i=i: inter.__setitem__(f"layer{i}", o[0].detach())))
    if ec["do_mask_input"]:
        hs.append(enc.masking.register_forward_hook(lambda m, i, o: masks.append(o[1].numpy().copy())))
    cut = (lambda a: a) if full else (lambda a: a[:, ::max(1, a.shape[1] // 16), ::7, ::13])   # (B, C, P, D): every C/16-th channel too

    def record(tag, out):
        fx[tag + "_loss"] = out.loss.detach().numpy(); fx[tag + "_n_examples"] = out.n_examples.numpy()
        fx[tag + "_preds"] = out.preds.detach().numpy() if full else out.preds.detach().numpy()[..., ::3, :]
        if method == "mlm":
            fx[tag + "_mask"] = out.mask.numpy(); fx[tag + "_raw_mask"] = masks[-1]
        fx[tag + "_embed"] = cut(inter["embed"].numpy())
        for i in range(len(enc.encoder.layers)):
            fx[f"{tag}_layer{i}"] = cut(inter[f"layer{i}"].numpy())

    model.eval()
    with torch.no_grad():
        out = model(**{k: v.clone() for k, v in batch.items()})
    record("eval0", out)
    if method == "mlm":
        fx["patch_input"] = out.patch_input.numpy() if full else out.patch_input.numpy()[:, ::3]
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=5e-5, eps=1e-8)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, total_steps=100, max_lr=1e-3, pct_start=0.0, div_factor=25)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for s in range(steps):
        out = model(**{k: v.clone() for k, v in batch.items()})
        out.loss.backward()
        record(f"step{s}", out)
        if s == 0:
            for k, p in model.named_parameters():
                if p.grad is None:
                    continue
                if full:
                    fx["grad:" + k] = p.grad.numpy().copy()
                else:
                    sm = summarise(p.grad)
                    fx["gsum:" + k] = np.array([sm["sum"], sm["abssum"]]); fx["gidx:" + k], fx["gval:" + k] = sm["idx"], sm["val"]
        opt.step(); sched.step(); opt.zero_grad()
    model.eval()
    with torch.no_grad():
        out = model(**{k: v.clone() for k, v in batch.items()})
    record("eval2", out)      # uses the BatchNorm running statistics of the two train steps
    for k, v in model.state_dict().items():
        v = v.float() if v.dtype == torch.int64 else v
        if full:
            fx["w0:" + k] = sd0[k].float().numpy(); fx["w2:" + k] = v.numpy()
        else:
            a, b = summarise(sd0[k].float()), summarise(v)
            fx["w0sum:" + k] = np.array([a["sum"], a["abssum"]]); fx["w0idx:" + k], fx["w0val:" + k] = a["idx"], a["val"]
            fx["w2val:" + k] = b["val"]
    fx["config_json"] = np.array(json.dumps(over)); fx["lens"] = np.array(lens)
    fx["kwargs_json"] = np.array(json.dumps({k: v for k, v in kw.items()}))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
    print(name, "eval loss", float(fx["eval0_loss"]), "train loss", float(fx["step0_loss"]), "n", int(fx["step0_n_examples"]),
          "params", sum(p.numel() for p in model.parameters()), list(sd0)[:4], out.preds.shape)


def ptst_tiny(**enc_extra):
    enc = {"num_input_channels": 6, "context_length": 45, "patch_length": 10, "patch_stride": 10, "num_hidden_layers": 2, "d_model": 32,
           "num_attention_heads": 2, "ffn_dim": 64, "attention_dropout": 0.0, "ff_dropout": 0.0, "do_mask_input": False}
    enc.update(enc_extra)
    return {"encoder": enc}


if __name__ == "__main__" and "--ptst" in sys.argv:
    ptst_case("g_ptst_tiny", ptst_tiny(), "ctc", 3, [45, 38, 30], [2, 2, 1])
    ptst_case("g_ptst_tiny_ov", {**ptst_tiny(patch_stride=5, context_length=48), "decoder": {"mlp_decoder": True}}, "ctc", 3, [48, 40, 29], [4, 3, 2])
    ptst_case("g_ptst_tiny_mlm", ptst_tiny(do_mask_input=True, random_mask_ratio=0.4), "mlm", 3, [45, 38, 30])
    ptst_case("g_ptst_tiny_mlm_rate", ptst_tiny(do_mask_input=True, random_mask_ratio=0.4), "mlm", 3, [45, 38, 30], log_input=False)
    ptst_case("g_ptst_c5", {"encoder": {"num_input_channels": 128, "context_length": 2050, "attention_dropout": 0.0, "ff_dropout": 0.0,
                                         "do_mask_input": False}}, "ctc", 2, [2050, 1500], [60, 40], vocab=41, full=False)


# ------------------------------------------------------------------------------------------------
# checkpoint interchange — `--ckpt`
#   forward : the REFERENCE's save_checkpoint (ndt1.py:685-688, bci.py:250-257, itransformer.py:403-407, patchtst.py:258-262) on tiny
#             configs -> tests/golden/ckpt_<model>/ (the files it wrote, nothing else) + expected.npz (inputs, eval outputs, checksums of
#             every state-dict tensor). GPU / CPU tests load these directories with the native load_checkpoint / from_pt.
#   reverse : a NATIVE model built on the CPU writes its checkpoint to a temp dir; the REFERENCE loads it (from_pt / load_checkpoint) and
#             every state-dict tensor must be equal. Build container only (needs the reference); the outcome is recorded in
#             tests/golden/ckpt_reverse_check.json.
# ------------------------------------------------------------------------------------------------
def _sd_summary(sd):
    return {k: [list(v.shape), float(v.double().sum()), float(v.double().abs().sum())] for k, v in sd.items()}


def ckpt_cases():
    import shutil
    import tempfile
    import types
    REPO = os.path.dirname(os.path.dirname(OUT))
    sys.path.insert(0, REPO)
    peft = types.ModuleType("peft"); peft.LoraConfig = None; peft.get_peft_model = None
    sys.modules.setdefault("peft", peft)
    _install_torchvision_mlp()
    from transformers import AutoModelForCausalLM, LlamaConfig
    from models.bci import BCI
    from models.itransformer import iTransformer
    from models.patchtst import PatchTSTForSpikingActivity
    if sys.modules.get("peft") is peft:
        del sys.modules["peft"]       # (the import stand-in: transformers' from_pretrained probes for the real package)
    reverse = {}

    def fresh(d):
        p = os.path.join(OUT, d)
        shutil.rmtree(p, ignore_errors=True)
        os.makedirs(p)
        return p

    def same(a, b):
        assert a.keys() == b.keys(), (sorted(a.keys() ^ b.keys()))
        for k in a:
            assert torch.equal(a[k].float().cpu(), b[k].float().cpu()), k
        return len(a)

    # ---------------- NDT1 (ctc, tiny) ----------------
    over = det(tiny())
    cfg = update_config("configs/ndt1.yaml", over)
    torch.manual_seed(7)
    ref = NDT1(cfg, method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True)
    d = fresh("ckpt_ndt1")
    ref.save_checkpoint(d)
    rows, batch, _ = make_batch(ref, [40, 33, 21], 16, [5, 4, 2], 11, seed=3)
    ref.eval()
    with torch.no_grad():
        out = ref(**batch)
    fx = {"in_" + k: v.numpy() for k, v in batch.items()}
    fx.update(preds=out.preds.numpy(), loss=out.loss.numpy(), config_json=np.array(json.dumps(over)),
              state_json=np.array(json.dumps(_sd_summary(ref.state_dict()))))
    np.savez_compressed(os.path.join(d, "expected.npz"), **fx)
    # reverse
    from llm_bci_amd.ndt1 import NDT1 as NNDT1
    torch.manual_seed(11)
    nat = NNDT1(json.loads(json.dumps(over)), method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True, compute_dtype="fp32")
    with tempfile.TemporaryDirectory() as td:
        nat.save_checkpoint(td)
        o2 = json.loads(json.dumps(over)); o2["encoder"]["from_pt"] = td
        r2 = NDT1(update_config("configs/ndt1.yaml", o2), method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True)   # ndt1.py:468-476,503-504
        n = same(dict(nat.state_dict()), dict(r2.state_dict()))
        r3 = NDT1(cfg, method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True)
        r3.load_checkpoint(td)
        same(dict(nat.state_dict()), dict(r3.state_dict()))
    reverse["NDT1"] = {"reference_from_pt_equal_tensors": n, "reference_load_checkpoint": "equal"}

    # ---------------- iTransformer (mlm, tiny) ----------------
    iover = itr_tiny()
    icfg = update_config("configs/itransformer.yaml", iover)
    torch.manual_seed(7)
    iref = iTransformer(icfg, method_name="mlm", log_input=True, loss="poisson_nll")
    d = fresh("ckpt_itransformer")
    iref.save_checkpoint(d)
    np.savez_compressed(os.path.join(d, "expected.npz"), config_json=np.array(json.dumps(iover)),
                        state_json=np.array(json.dumps(_sd_summary({k: v for k, v in iref.state_dict().items() if not k.startswith("masker")}))))
    from llm_bci_amd.itransformer import iTransformer as NITR
    torch.manual_seed(11)
    inat = NITR(json.loads(json.dumps(iover)), method_name="mlm", log_input=True, loss="poisson_nll", compute_dtype="fp32")
    with tempfile.TemporaryDirectory() as td:
        inat.save_checkpoint(td)
        o2 = json.loads(json.dumps(iover)); o2["encoder"]["from_pt"] = td; o2.setdefault("decoder", {})["from_pt"] = td
        r2 = iTransformer(update_config("configs/itransformer.yaml", o2), method_name="mlm", log_input=True, loss="poisson_nll")   # itransformer.py:226-250
        n = same({k: v for k, v in inat.state_dict().items()}, {k: v for k, v in r2.state_dict().items() if not k.startswith("masker")})
    reverse["iTransformer"] = {"reference_from_pt_equal_tensors": n}

    # ---------------- PatchTST (ctc, tiny) ----------------
    pover = ptst_tiny()
    pcfg = update_config("configs/patchtst.yaml", pover)
    torch.manual_seed(7)
    pref = PatchTSTForSpikingActivity(pcfg, method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True)
    d = fresh("ckpt_patchtst")
    pref.save_checkpoint(d)
    g = np.random.default_rng(5)
    B, T, Cn = 3, 45, 6
    spikes = g.standard_normal((B, T, Cn)).astype(np.float32); smask = np.zeros((B, T), np.int64)
    for b, L in enumerate([45, 38, 30]):
        spikes[b, L:] = 0; smask[b, :L] = 1
    tg = np.zeros((B, 2), np.int64)
    for b, n_t in enumerate([2, 2, 1]):
        tg[b, :n_t] = g.integers(1, 11, n_t)
    pb = {"spikes": torch.from_numpy(spikes), "spikes_mask": torch.from_numpy(smask), "spikes_lengths": torch.tensor([45, 38, 30]),
          "targets": torch.from_numpy(tg), "targets_lengths": torch.tensor([2, 2, 1])}   # (the reference's ctc forward needs targets, patchtst.py:248)
    pref.eval()
    with torch.no_grad():
        pout = pref(**pb)
    fx = {"in_" + k: v.numpy() for k, v in pb.items()}
    fx.update(preds=pout.preds.numpy(), loss=pout.loss.numpy(), config_json=np.array(json.dumps(pover)), state_json=np.array(json.dumps(_sd_summary(pref.state_dict()))))
    np.savez_compressed(os.path.join(d, "expected.npz"), **fx)
    from llm_bci_amd.patchtst import PatchTSTForSpikingActivity as NPT
    torch.manual_seed(11)
    pnat = NPT(json.loads(json.dumps(pover)), method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True, compute_dtype="fp32")
    with tempfile.TemporaryDirectory() as td:
        pnat.save_checkpoint(td)
        r3 = PatchTSTForSpikingActivity(pcfg, method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True)
        r3.load_checkpoint(td)                                         # patchtst.py:264-266 (its from_pt reads the pickled config as yaml and cannot work)
        n = same(dict(pnat.state_dict()), dict(r3.state_dict()))
    reverse["PatchTST"] = {"reference_load_checkpoint_equal_tensors": n,
                           "note": "reference from_pt (patchtst.py:174-176) parses encoder_config.yaml as yaml although save_checkpoint pickles it: not usable in the reference itself"}

    # ---------------- BCI (tiny Llama through the reference's llm= argument) ----------------
    enc = json.loads(json.dumps(det(tiny())["encoder"]))
    bcfg = {"projector": {"stacking": 2, "inter_size": 48, "bias": True, "act": "relu"}, "ndt1": {"encoder": enc}}
    lcfg = dict(vocab_size=128, hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
                max_position_embeddings=128)
    torch.manual_seed(7)
    llm = AutoModelForCausalLM.from_config(LlamaConfig(**lcfg))
    bref = BCI(json.loads(json.dumps(bcfg)), llm_path=None, llm=llm, method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True)
    d = fresh("ckpt_bci")
    bref.save_checkpoint(d)
    np.savez_compressed(os.path.join(d, "expected.npz"), config_json=np.array(json.dumps(bcfg)), llm_config_json=np.array(json.dumps(lcfg)),
                        ndt1_state_json=np.array(json.dumps(_sd_summary(bref.ndt1.state_dict()))),
                        projector_state_json=np.array(json.dumps(_sd_summary(bref.projector.state_dict()))),
                        llm_state_json=np.array(json.dumps(_sd_summary(bref.llm.state_dict()))))
    from llm_bci_amd.bci import BCI as NBCI
    torch.manual_seed(11)
    nllm = AutoModelForCausalLM.from_config(LlamaConfig(**lcfg))
    bnat = NBCI(json.loads(json.dumps(bcfg)), llm=nllm, method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True, compute_dtype="fp32")
    with tempfile.TemporaryDirectory() as td:
        bnat.save_checkpoint(td)
        o2 = json.loads(json.dumps(bcfg)); o2["from_pt"] = td
        r2 = BCI(o2, llm_path=None, method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True)      # bci.py:46,57,78-80,99-104
        n = same(dict(bnat.ndt1.state_dict()), dict(r2.ndt1.state_dict())) + same(dict(bnat.projector.state_dict()), dict(r2.projector.state_dict()))
        n += same({k: v for k, v in bnat.llm.state_dict().items()}, {k: v for k, v in r2.llm.state_dict().items()})
    reverse["BCI"] = {"reference_from_pt_equal_tensors": n}
    json.dump({"made_by": "tests/golden/make_golden.py --ckpt (build container, reference imported from /root/reference)",
               "native_checkpoint_loaded_by_reference": reverse}, open(os.path.join(OUT, "ckpt_reverse_check.json"), "w"), indent=2, sort_keys=True)
    for dd in ("ckpt_ndt1", "ckpt_itransformer", "ckpt_patchtst", "ckpt_bci"):
        print(dd, sorted(os.listdir(os.path.join(OUT, dd))))
    print("reverse:", reverse)


def ckpt_itr_cases():
    """iTransformer checkpoints with every optional table / the other embedder (round 4): region + depth embeddings under the mlp embedder,
    and the UnivariateTransformer embedder. Forward: written by the REFERENCE's save_checkpoint into tests/golden/ckpt_itransformer_{full,uni}/
    with an eval forward (mask replayed) recorded; reverse: a native checkpoint loaded by the reference's from_pt, tensors equal - the record
    is merged into ckpt_reverse_check.json."""
    import shutil
    import tempfile
    REPO = os.path.dirname(os.path.dirname(OUT))
    sys.path.insert(0, REPO)
    _install_torchvision_mlp()
    from models.itransformer import iTransformer
    from llm_bci_amd.itransformer import iTransformer as NITR
    rec = json.load(open(os.path.join(OUT, "ckpt_reverse_check.json")))
    emb = {"mode": "transformer", "max_n_bins": 12, "dropout": 0.0, "hidden_size": 16, "n_heads": 2, "n_layers": 2, "activation": "relu"}
    cases = {"ckpt_itransformer_full": itr_tiny(embed_region=True, regions=["CA1", "DG", "LP", "PO"], embed_depth=True),
             "ckpt_itransformer_uni": itr_tiny(embedder=emb, embed_region=True, regions=["CA1", "DG"], embed_depth=False)}
    for dname, iover in cases.items():
        icfg = update_config("configs/itransformer.yaml", iover)
        torch.manual_seed(7)
        iref = iTransformer(icfg, method_name="mlm", log_input=True, loss="poisson_nll")
        d = os.path.join(OUT, dname)
        shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
        iref.save_checkpoint(d)
        g = np.random.default_rng(3)
        B, N, T = 3, 10, 12
        spikes = g.poisson(0.5, (B, T, N)).astype(np.float32)
        regs = list(iover["encoder"]["regions"])
        nr = np.array(regs)[g.integers(0, len(regs), (B, N))]
        batch = {"spikes": torch.from_numpy(spikes), "spikes_mask": torch.ones(B, T, dtype=torch.int64),
                 "spikes_timestamp": torch.arange(T).repeat(B, 1), "neuron_regions": nr}
        fx = {"in_spikes": spikes, "in_spikes_mask": np.ones((B, T), np.int64), "in_spikes_timestamp": np.tile(np.arange(T), (B, 1)),
              "in_neuron_regions": nr.astype("U16")}
        if iover["encoder"].get("embed_depth"):
            nd = g.uniform(0, 3.84, (B, N)).astype(np.float32)
            batch["neuron_depths"] = torch.from_numpy(nd); fx["in_neuron_depths"] = nd
        masks = []
        for mk in iref.masker.values():
            mk.register_forward_hook(lambda mod, inp, out: masks.append(out[1].numpy().copy()))
        iref.eval()
        with torch.no_grad():
            out = iref(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
        fx.update(raw_mask=masks[-1], preds=out.preds.numpy(), loss=out.loss.numpy(), n_examples=out.n_examples.numpy(),
                  config_json=np.array(json.dumps(iover)),
                  state_json=np.array(json.dumps(_sd_summary({k: v for k, v in iref.state_dict().items() if not k.startswith("masker")}))))
        np.savez_compressed(os.path.join(d, "expected.npz"), **fx)
        torch.manual_seed(11)
        inat = NITR(json.loads(json.dumps(iover)), method_name="mlm", log_input=True, loss="poisson_nll", compute_dtype="fp32")
        with tempfile.TemporaryDirectory() as td:
            inat.save_checkpoint(td)
            o2 = json.loads(json.dumps(iover)); o2["encoder"]["from_pt"] = td; o2.setdefault("decoder", {})["from_pt"] = td
            r2 = iTransformer(update_config("configs/itransformer.yaml", o2), method_name="mlm", log_input=True, loss="poisson_nll")   # itransformer.py:226-250
            a = {k: v for k, v in inat.state_dict().items()}
            b = {k: v for k, v in r2.state_dict().items() if not k.startswith("masker")}
            assert a.keys() == b.keys(), sorted(a.keys() ^ b.keys())
            for k in a:
                assert torch.equal(a[k].float().cpu(), b[k].float().cpu()), k
        rec["native_checkpoint_loaded_by_reference"]["iTransformer_" + dname.rsplit("_", 1)[1]] = {"reference_from_pt_equal_tensors": len(a)}
        print(dname, sorted(os.listdir(d)), "reverse tensors", len(a), "loss", float(out.loss))
    json.dump(rec, open(os.path.join(OUT, "ckpt_reverse_check.json"), "w"), indent=2, sort_keys=True)


if __name__ == "__main__" and "--ckpt-itr" in sys.argv:
    ckpt_itr_cases()

if __name__ == "__main__" and "--ckpt" in sys.argv:
    ckpt_cases()
