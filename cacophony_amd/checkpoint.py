"""Checkpoint ingestion.

Two containers hold Cacophony weights:

* torch: `torch.load(path)` -> state dict, possibly wrapped in `model_state_dict` / `state_dict`
  (src/eval/eval_caco_torch.py:154-169).  Its keys are what `CACO.load_state_dict` / `caco_load_tensor` take.
* JAX / Flax: `flax.training.checkpoints.restore_checkpoint(path, target=None)['0']['params']`
  (src/caco/load_model.py:12-16, :65-69): a msgpack file holding the Flax parameter TREE.  Neither jax nor flax exist
  here, so this module reads the container itself (flax.serialization's published msgpack layout: ndarray =
  ExtType(1, msgpack((shape, dtype name, raw bytes)))) and maps the tree onto the torch key names.

The tree layout below is derived from the module definitions of the JAX twin (no checkpoint is available offline, so it
is pinned by a synthetic round trip - tests/test_checkpoint.py - not by a real file; the names follow Flax's
auto-numbering of `nn.compact` submodules in creation order and the explicit attribute names of `setup` modules):

  audio_module/                                         src/caco/audio_models/mae.py:107-141
      Dense_0/{kernel [256,H], bias}                    -> audio_module.input_proj.{weight = kernel^T, bias}
      freq_positional_embedding [8,H]                   -> audio_module.freq_positional_embedding
      AudioEncoderLayer_{i}/                            :72-98
          LayerNorm_0, LayerNorm_1 /{scale, bias}       -> layers.i.norm1 / norm2 .{weight, bias}
          MultiHeadDotProductAttention_0/
              query|key|value /{kernel [H,heads,hd], bias [heads,hd]}
                                                        -> layers.i.attn.in_proj_weight = [Wq; Wk; Wv], W* = kernel.reshape(H, H)^T
              out/{kernel [heads,hd,H], bias [H]}       -> layers.i.attn.out_proj.{weight = kernel.reshape(H, H)^T, bias}
          MLP_0/Dense_0, Dense_1                        -> layers.i.mlp.fc1 / fc2           :55-69
      LayerNorm_0                                       -> audio_module.norm
  audio_attention_pool/                                 src/caco/caco.py:19-53
      Dense_0 (keys | values), query [H], Dense_1       -> kv_proj, query, out_proj
  text_proj/{kernel, bias}, logit_scale                 caco.py:63-70
  text_module/ | decoder_module/                        src/caco/text_models/roberta_text_model.py:539-612
      embeddings/{word,position,token_type}_embeddings/embedding, embeddings/LayerNorm      (text_module only)
      encoder/layer/ScanFlaxRobertaLayer_0/...  every leaf STACKED over the layers on axis 0 (nn.scan, :449-455), or
      encoder/layer/{i}/...                      the unrolled form (:464-465)                -> encoder.layers.i....
          attention/self/{query,key,value}, attention/output/{dense, LayerNorm}, intermediate/dense,
          output/{dense, LayerNorm}, crossattention/...  (decoder_module only)
      pooler/{key_proj, value_proj, attention_pool_query}                                    (text_module only)
      decoder_proj/{kernel, bias}                                                            (decoder_module only)

Every Flax `Dense` kernel is [in, out]: the torch weight is its transpose.  Differences between the two reference
implementations that a converted checkpoint does NOT carry and the caller must set in the config:
the JAX model pools with 8 heads (load_model.py:46) where create_caco_model defaults to 2, and Flax LayerNorm defaults
to eps 1e-6 in the audio stack where torch uses 1e-5.
"""
from __future__ import annotations

import re
from typing import Dict, Mapping, MutableMapping, Optional, Tuple

import numpy as np

Tree = Mapping[str, object]


# ---------------------------------------------------------------------------------------------------------------
# Flax msgpack container (flax.serialization: msgpack_restore / msgpack_serialize)
# ---------------------------------------------------------------------------------------------------------------
def _ext_hook(code: int, data: bytes):
    import msgpack
    if code == 1:        # ndarray: (shape, dtype name, buffer)
        shape, dtype, buf = msgpack.unpackb(data, raw=False, strict_map_key=False)
        return np.frombuffer(buf, dtype=np.dtype(dtype)).reshape(shape).copy()
    if code == 2:        # native complex
        re_, im_ = msgpack.unpackb(data, raw=False)
        return complex(re_, im_)
    if code == 3:        # numpy scalar, stored like a 0-d ndarray
        shape, dtype, buf = msgpack.unpackb(data, raw=False, strict_map_key=False)
        return np.frombuffer(buf, dtype=np.dtype(dtype)).reshape(shape)[()]
    import msgpack as _m
    return _m.ExtType(code, data)


def read_flax_msgpack(path: str) -> Tree:
    """The state tree of a Flax checkpoint file (what `restore_checkpoint(path, target=None)` returns)."""
    import msgpack
    with open(path, "rb") as f:
        return msgpack.unpackb(f.read(), ext_hook=_ext_hook, raw=False, strict_map_key=False)


def write_flax_msgpack(tree: Tree, path: str) -> None:
    """Inverse of read_flax_msgpack (used by the round-trip test and to hand weights to the JAX twin)."""
    import msgpack

    def default(o):
        if isinstance(o, np.ndarray):
            a = np.ascontiguousarray(o)
            return msgpack.ExtType(1, msgpack.packb((list(a.shape), a.dtype.name, a.tobytes()), use_bin_type=True))
        if isinstance(o, np.generic):
            a = np.asarray(o)
            return msgpack.ExtType(3, msgpack.packb((list(a.shape), a.dtype.name, a.tobytes()), use_bin_type=True))
        raise TypeError(f"cannot serialise {type(o)}")

    with open(path, "wb") as f:
        f.write(msgpack.packb(tree, default=default, use_bin_type=True))


def flax_params(state_tree: Tree) -> Tree:
    """`caco_state_dict['0']['params']` of load_model.py:15-16 (accepts the params tree itself as well)."""
    t = state_tree
    for key in ("0", 0):
        if isinstance(t, Mapping) and key in t:
            t = t[key]
            break
    if isinstance(t, Mapping) and "params" in t:
        t = t["params"]
    return t


# ---------------------------------------------------------------------------------------------------------------
# tree <-> state dict
# ---------------------------------------------------------------------------------------------------------------
def _f32(x) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(x), dtype=np.float32)


def _dense(out: MutableMapping[str, np.ndarray], key: str, node: Mapping) -> None:
    out[key + ".weight"] = _f32(np.asarray(node["kernel"]).T)
    out[key + ".bias"] = _f32(node["bias"])


def _ln(out, key, node) -> None:
    out[key + ".weight"] = _f32(node["scale"])
    out[key + ".bias"] = _f32(node["bias"])


_ROBERTA_DENSE = ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense",
                  "intermediate.dense", "output.dense", "crossattention.self.query", "crossattention.self.key",
                  "crossattention.self.value", "crossattention.output.dense")
_ROBERTA_LN = ("attention.output.LayerNorm", "output.LayerNorm", "crossattention.output.LayerNorm")


def _get(node: Mapping, dotted: str) -> Optional[Mapping]:
    for part in dotted.split("."):
        if not isinstance(node, Mapping) or part not in node:
            return None
        node = node[part]
    return node


def _roberta_layers(out, prefix: str, layer_node: Mapping) -> int:
    """encoder/layer: scan-stacked (leading layer axis) or unrolled ('0', '1', ...) -> `prefix.encoder.layers.i.*`."""
    if "ScanFlaxRobertaLayer_0" in layer_node:
        stacked = layer_node["ScanFlaxRobertaLayer_0"]
        n = int(np.asarray(_get(stacked, "attention.self.query")["kernel"]).shape[0])
        per_layer = [None] * n

        def pick(node, i):
            return {k: (pick(v, i) if isinstance(v, Mapping) else np.asarray(v)[i]) for k, v in node.items()}
        per_layer = [pick(stacked, i) for i in range(n)]
    else:
        idx = sorted(int(k) for k in layer_node.keys() if re.fullmatch(r"\d+", str(k)))
        per_layer = [layer_node[str(i)] if str(i) in layer_node else layer_node[i] for i in idx]
    for i, L in enumerate(per_layer):
        base = f"{prefix}.encoder.layers.{i}"
        for name in _ROBERTA_DENSE:
            node = _get(L, name)
            if node is not None:
                _dense(out, f"{base}.{name}", node)
        for name in _ROBERTA_LN:
            node = _get(L, name)
            if node is not None:
                _ln(out, f"{base}.{name}", node)
    return len(per_layer)


def flax_to_state_dict(params: Tree) -> Dict[str, np.ndarray]:
    """Flax parameter tree of the JAX CACO (or of an AudioEncoder alone) -> the torch state dict `CACO.load_state_dict`
    takes.  Sub-trees that are absent (no text tower, no decoder) are skipped; unknown top-level keys raise."""
    params = flax_params(params)
    known = {"audio_module", "audio_attention_pool", "text_module", "decoder_module", "text_proj", "logit_scale"}
    if "Dense_0" in params and "audio_module" not in params:      # an AudioEncoder tree on its own (load_audiomae, :65-69)
        params = {"audio_module": params}
    unknown = set(params.keys()) - known
    if unknown:
        raise ValueError(f"flax_to_state_dict: unknown top-level entries {sorted(unknown)}")
    out: Dict[str, np.ndarray] = {}
    if "logit_scale" in params:
        out["logit_scale"] = _f32(params["logit_scale"]).reshape(())
    if "audio_module" in params:
        a = params["audio_module"]
        _dense(out, "audio_module.input_proj", a["Dense_0"])
        out["audio_module.freq_positional_embedding"] = _f32(a["freq_positional_embedding"])
        n = 0
        while f"AudioEncoderLayer_{n}" in a:
            L, base = a[f"AudioEncoderLayer_{n}"], f"audio_module.layers.{n}"
            _ln(out, base + ".norm1", L["LayerNorm_0"])
            _ln(out, base + ".norm2", L["LayerNorm_1"])
            att = L["MultiHeadDotProductAttention_0"]
            H = int(np.asarray(att["query"]["kernel"]).shape[0])
            w = [np.asarray(att[k]["kernel"]).reshape(H, -1).T for k in ("query", "key", "value")]      # [heads*hd, H] each
            b = [np.asarray(att[k]["bias"]).reshape(-1) for k in ("query", "key", "value")]
            out[base + ".attn.in_proj_weight"] = _f32(np.concatenate(w, 0))
            out[base + ".attn.in_proj_bias"] = _f32(np.concatenate(b, 0))
            out[base + ".attn.out_proj.weight"] = _f32(np.asarray(att["out"]["kernel"]).reshape(-1, H).T)
            out[base + ".attn.out_proj.bias"] = _f32(att["out"]["bias"])
            _dense(out, base + ".mlp.fc1", L["MLP_0"]["Dense_0"])
            _dense(out, base + ".mlp.fc2", L["MLP_0"]["Dense_1"])
            n += 1
        if n == 0:
            raise ValueError("flax_to_state_dict: audio_module holds no AudioEncoderLayer_i")
        _ln(out, "audio_module.norm", a["LayerNorm_0"])
    if "audio_attention_pool" in params:
        p = params["audio_attention_pool"]
        _dense(out, "audio_attention_pool.kv_proj", p["Dense_0"])
        out["audio_attention_pool.query"] = _f32(p["query"])
        _dense(out, "audio_attention_pool.out_proj", p["Dense_1"])
    if "text_proj" in params:
        _dense(out, "text_proj", params["text_proj"])
    if "text_module" in params:
        t = params["text_module"]
        e = t["embeddings"]
        for nm in ("word_embeddings", "position_embeddings", "token_type_embeddings"):
            out[f"text_module.embeddings.{nm}.weight"] = _f32(e[nm]["embedding"])
        _ln(out, "text_module.embeddings.LayerNorm", e["LayerNorm"])
        _roberta_layers(out, "text_module", t["encoder"]["layer"])
        pl = t["pooler"]
        out["text_module.pooler.attention_pool_query"] = _f32(pl["attention_pool_query"])
        _dense(out, "text_module.pooler.key_proj", pl["key_proj"])
        _dense(out, "text_module.pooler.value_proj", pl["value_proj"])
    if "decoder_module" in params:
        d = params["decoder_module"]
        _roberta_layers(out, "decoder_module", d["encoder"]["layer"])
        _dense(out, "decoder_module.decoder_proj", d["decoder_proj"])
    return out


def state_dict_to_flax(state: Mapping[str, object], audio_heads: int = 8, scan: bool = True) -> Dict[str, object]:
    """Inverse mapping (torch state dict -> Flax parameter tree): documents the layout and feeds the round-trip test."""
    sd = {k: np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v, dtype=np.float32) for k, v in state.items()}

    def dense(key):
        return {"kernel": sd[key + ".weight"].T.copy(), "bias": sd[key + ".bias"].copy()}

    def ln(key):
        return {"scale": sd[key + ".weight"].copy(), "bias": sd[key + ".bias"].copy()}

    tree: Dict[str, object] = {}
    if "logit_scale" in sd:
        tree["logit_scale"] = sd["logit_scale"].reshape(())
    if "audio_module.input_proj.weight" in sd:
        a: Dict[str, object] = {"Dense_0": dense("audio_module.input_proj"),
                                "freq_positional_embedding": sd["audio_module.freq_positional_embedding"].copy(),
                                "LayerNorm_0": ln("audio_module.norm")}
        n = 0
        while f"audio_module.layers.{n}.norm1.weight" in sd:
            base = f"audio_module.layers.{n}"
            W, B = sd[base + ".attn.in_proj_weight"], sd[base + ".attn.in_proj_bias"]
            H = W.shape[1]
            hd = H // audio_heads
            att = {}
            for j, name in enumerate(("query", "key", "value")):
                att[name] = {"kernel": W[j * H:(j + 1) * H].T.reshape(H, audio_heads, hd).copy(),
                             "bias": B[j * H:(j + 1) * H].reshape(audio_heads, hd).copy()}
            att["out"] = {"kernel": sd[base + ".attn.out_proj.weight"].T.reshape(audio_heads, hd, H).copy(),
                          "bias": sd[base + ".attn.out_proj.bias"].copy()}
            a[f"AudioEncoderLayer_{n}"] = {"LayerNorm_0": ln(base + ".norm1"), "LayerNorm_1": ln(base + ".norm2"),
                                          "MultiHeadDotProductAttention_0": att,
                                          "MLP_0": {"Dense_0": dense(base + ".mlp.fc1"), "Dense_1": dense(base + ".mlp.fc2")}}
            n += 1
        tree["audio_module"] = a
    if "audio_attention_pool.query" in sd:
        tree["audio_attention_pool"] = {"Dense_0": dense("audio_attention_pool.kv_proj"), "query": sd["audio_attention_pool.query"].copy(),
                                        "Dense_1": dense("audio_attention_pool.out_proj")}
    if "text_proj.weight" in sd:
        tree["text_proj"] = dense("text_proj")

    def roberta_layers(prefix):
        layers = []
        i = 0
        while f"{prefix}.encoder.layers.{i}.attention.self.query.weight" in sd:
            base = f"{prefix}.encoder.layers.{i}"
            L: Dict[str, object] = {}
            for name in _ROBERTA_DENSE + _ROBERTA_LN:
                k = f"{base}.{name}"
                if k + ".weight" not in sd:
                    continue
                node = L
                parts = name.split(".")
                for part in parts[:-1]:
                    node = node.setdefault(part, {})
                node[parts[-1]] = ln(k) if name in _ROBERTA_LN else dense(k)
            layers.append(L)
            i += 1
        if not scan:
            return {str(i): L for i, L in enumerate(layers)}

        def stack(nodes):
            return {k: (stack([nd[k] for nd in nodes]) if isinstance(v, dict) else np.stack([nd[k] for nd in nodes], 0))
                    for k, v in nodes[0].items()}
        return {"ScanFlaxRobertaLayer_0": stack(layers)}

    if "text_module.embeddings.word_embeddings.weight" in sd:
        tree["text_module"] = {
            "embeddings": {**{nm: {"embedding": sd[f"text_module.embeddings.{nm}.weight"].copy()}
                              for nm in ("word_embeddings", "position_embeddings", "token_type_embeddings")},
                           "LayerNorm": ln("text_module.embeddings.LayerNorm")},
            "encoder": {"layer": roberta_layers("text_module")},
            "pooler": {"attention_pool_query": sd["text_module.pooler.attention_pool_query"].copy(),
                       "key_proj": dense("text_module.pooler.key_proj"), "value_proj": dense("text_module.pooler.value_proj")}}
    if "decoder_module.decoder_proj.weight" in sd:
        tree["decoder_module"] = {"encoder": {"layer": roberta_layers("decoder_module")},
                                  "decoder_proj": dense("decoder_module.decoder_proj")}
    return tree


def checkpoint_format(path: str) -> str:
    """'torch' for a torch.save file (zip container, or the legacy pickle stream: 0x80 + protocol 2..5), 'flax' otherwise
    (a msgpack map starts 0x81..0x8f / 0xde / 0xdf).  The container is sniffed, not guessed from a failed load."""
    with open(path, "rb") as f:
        head = f.read(4)
    if head[:4] == b"PK\x03\x04" or (len(head) >= 2 and head[0] == 0x80 and 2 <= head[1] <= 5):
        return "torch"
    return "flax"


def load_checkpoint(path: str, weights_only: bool = True) -> Dict[str, np.ndarray]:
    """Either container -> state dict with the torch key names.  torch files (`checkpoint_format`) go through torch.load -
    its errors surface as they are; `weights_only=False` is the reference's plain torch.load (eval_caco_torch.py:158) for
    files that pickle more than tensors - and the `model_state_dict` / `state_dict` wrappers are unwrapped (:160-166);
    anything else is read as a Flax msgpack file."""
    if checkpoint_format(path) == "flax":
        return flax_to_state_dict(read_flax_msgpack(path))
    import torch
    obj = torch.load(path, map_location="cpu", weights_only=weights_only)
    for key in ("model_state_dict", "state_dict"):
        if isinstance(obj, Mapping) and key in obj and isinstance(obj[key], Mapping):
            obj = obj[key]
            break
    return {k: np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v) for k, v in obj.items()}
