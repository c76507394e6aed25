"""Config plumbing for the NDT1 plugin surface.

Mirrors the behaviour of the reference's utils/config_utils.py (attribute-access dict :6-12,
`include:<path>` expansion :20-30, recursive merge where non-dict leaves overwrite :36-52,
update_config(default, override) :59-75) so the reference's yaml files and override dicts are
consumed unchanged. Implementation is independent; the NDT1 defaults are built in (same values
as configs/ndt1.yaml) so no yaml has to travel with the package, and a cwd-relative
configs/ndt1.yaml is honoured when present, as in the reference (ndt1.py:17,464).
"""
import copy
import os

import yaml


class DictConfig(dict):
    """dict with attribute access; nested dicts are wrapped on read."""

    def __getattr__(self, key):
        try:
            val = self[key]
        except KeyError as e:
            raise AttributeError(key) from e
        return DictConfig(val) if isinstance(val, dict) and not isinstance(val, DictConfig) else val

    def __setattr__(self, key, value):
        self[key] = value


def _expand_includes(node):
    if isinstance(node, str) and node.startswith("include:"):
        with open(node.split(":", 1)[1], "r") as fh:
            node = yaml.safe_load(fh)
    if isinstance(node, dict):
        return {k: _expand_includes(v) for k, v in node.items()}
    return node


def _merge(base, over):
    if not isinstance(over, dict):
        return copy.deepcopy(over)
    out = dict(base) if isinstance(base, dict) else {}
    for k, v in over.items():
        out[k] = _merge(out.get(k, {}), v)
    return out


def _load(x):
    if isinstance(x, str):
        with open(x, "r") as fh:
            return yaml.safe_load(fh)
    return copy.deepcopy(dict(x)) if x is not None else None


def update_config(default_config, config=None):
    """default <- config (recursively); either may be a yaml path; includes are expanded."""
    base = _expand_includes(_load(default_config))
    over = base if config is None else _expand_includes(_load(config))
    return DictConfig(_merge(base, over))


def ndt1_defaults():
    """Built-in NDT1 model defaults (values of the reference's configs/ndt1.yaml)."""
    masker = dict(active=False, mode="neuron", ratio=0.1, zero_ratio=1.0, random_ratio=1.0, expand_prob=0.0,
                  max_timespan=1, regions=None, channels=None)
    enc = dict(
        from_pt=None,
        masker=dict(neuron=masker),
        context=dict(forward=-2, backward=-2),
        smooth_and_noise=dict(noise=True, smooth_sd=2, white_noise_sd=1.0, constant_offset_sd=0.2),
        embedder=dict(n_channels=256, n_blocks=24, n_days=24, max_F=1024, input_dim=256, adapt=False, day_token=False,
                      block_token=False, pos=True, act="softsign", bias=True, dropout=0.2,
                      stack=dict(active=True, size=32, stride=4)),
        transformer=dict(n_layers=5, hidden_size=1024, use_rope=False, rope_theta=10000.0, n_heads=8,
                         attention_bias=True, act="gelu", inter_size=1024, mlp_bias=True, dropout=0.4, fixup_init=True),
        factors=dict(active=False, size=1024, act="relu", bias=True, dropout=0.0, fixup_init=False, init_range=0.1),
    )
    return dict(model_class="NDT1", encoder=enc, decoder=dict(from_pt=None))


def ndt1_config(config):
    """Model defaults merged with `config`, preferring a cwd-relative configs/ndt1.yaml if present."""
    base = "configs/ndt1.yaml" if os.path.exists("configs/ndt1.yaml") else ndt1_defaults()
    return update_config(base, config if config is not None else {})


def itransformer_defaults():
    """Built-in iTransformer model defaults (values of the reference's configs/itransformer.yaml)."""
    masker = dict(force_active=True, mode="neuron", ratio=0.1, zero_ratio=1.0, random_ratio=1.0, expand_prob=0.0, max_timespan=1,
                  channels=None, timesteps=None, mask_regions=None, target_regions=None, n_mask_regions=1)
    enc = dict(from_pt=None,
               embedder=dict(mode="mlp", activation="relu", dropout=0.2, n_heads=4, hidden_size=128, n_layers=4, max_n_bins=100),
               hidden_size=768, activation="relu", bias=True, dropout=0.4, n_heads=8, n_layers=5, max_n_channels=1500,
               embed_region=True, embed_depth=False, regions=None)
    return dict(model_class="iTransformer", masker=dict(main=masker), encoder=enc,
                decoder=dict(from_pt=None, mlp_decoder=True, activation="relu", use_cls=True))


def itransformer_config(config):
    base = "configs/itransformer.yaml" if os.path.exists("configs/itransformer.yaml") else itransformer_defaults()
    return update_config(base, config if config is not None else {})


def patchtst_defaults():
    """Built-in PatchTST model defaults (values of the reference's configs/patchtst.yaml)."""
    enc = dict(from_pt=None, num_input_channels=128, context_length=45, patch_length=10, patch_stride=10, num_hidden_layers=4,
               d_model=256, num_attention_heads=8, share_embedding=True, channel_attention=False, ffn_dim=1024, norm_type="batchnorm",
               norm_eps=1.0e-5, attention_dropout=0.4, positional_dropout=0.0, path_dropout=0.0, ff_dropout=0.4, bias=True,
               activation_function="gelu", pre_norm=True, positional_encoding_type="sincos", init_std=0.02, scaling=None,
               do_mask_input=True, mask_type="random", random_mask_ratio=0.1, channel_consistent_masking=False, mask_value=0)
    dec = dict(from_pt=None, share_projection=True, pooling_type="mean", head_dropout=0.0, mlp_decoder=False, mlp_activation="gelu")
    return dict(model_class="PatchTST", encoder=enc, decoder=dec)


def patchtst_config(config):
    base = "configs/patchtst.yaml" if os.path.exists("configs/patchtst.yaml") else patchtst_defaults()
    return update_config(base, config if config is not None else {})
