"""Initial weights of PatchTSTForSpikingActivity drawn exactly as the reference draws them (models/patchtst.py:176,190):
the encoder IS HF transformers' PatchTSTModel (a third-party dependency of the reference, patchtst.py:8), so its
constructor + `post_init()` are called here, on the CPU, purely as the initialiser (normal(0, init_std) Linear weights,
zero biases, the fixed sincos `position_enc`, BatchNorm buffers); the head Linears use nn.Linear's default init, in the
reference's order. With torch.manual_seed(s) beforehand the model equals the reference's bit for bit. No forward pass of
transformers is ever run by the product."""
import torch
import torch.nn as nn

ENC_DROP = ("from_pt",)


def reference_order_init(enc_cfg, dec_cfg, method, vocab_size=None, seed=None):
    from transformers import PatchTSTConfig, PatchTSTModel
    if seed is not None:
        torch.manual_seed(seed)
    ec = {k: v for k, v in dict(enc_cfg).items() if k not in ENC_DROP}
    enc = PatchTSTModel(PatchTSTConfig.from_dict(ec))
    out = {"encoder." + k: v.detach().clone() for k, v in enc.state_dict().items()}
    D, pl = ec["d_model"], ec["patch_length"]
    n_out = vocab_size if method == "ctc" else pl
    if not dec_cfg.get("share_projection", True):
        raise Exception("PatchTST HIP path supports decoder.share_projection: true")
    if dec_cfg.get("mlp_decoder", False):
        a, b = nn.Linear(D, D), nn.Linear(D, n_out)
        out.update({"decoder.projection.0.weight": a.weight.detach(), "decoder.projection.0.bias": a.bias.detach(),
                    "decoder.projection.2.weight": b.weight.detach(), "decoder.projection.2.bias": b.bias.detach()})
    else:
        a = nn.Linear(D, n_out)
        out.update({"decoder.projection.weight": a.weight.detach(), "decoder.projection.bias": a.bias.detach()})
    return out
