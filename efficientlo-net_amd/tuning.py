"""The configuration as a VALUE, not an environment.

Every choice of kernel FORM that is not a function of a call's arguments lives in one place:

  * the library's `elo_tuning` (include/elo.h: chain forms, row thresholds, narrow-set-conv form, ...): the C library reads no
    environment variable; `_lib.lib()` fills the struct ONCE from the ELO_* variables below when it loads the library;
  * the host-side choices of this package (which operators take a grouping pre-pass, the tile count from which select-k is LDS
    tiled, the launch-merging bound, ...), read from the environment ONCE, at import.

`snapshot()` is the whole tuning as a plain dict (bench.py prints it as `config.tuning`), `digest()` its hash:
model.PWCLONet.capture() records the digest and refuses to replay a graph under another tuning (a captured graph has the forms
of its capture baked in).  `override(...)` changes fields for a `with` block (tests, A/B measurements) and restores them.
All forms of an entry point compute the same function (bit for bit, or to fp32 summation order where stated): speed choices."""
import contextlib
import hashlib
import json
import os

# host-side fields: name -> (environment variable, parser, default)
def _prepass(spec):
    return None if spec in (None, "") else int(spec)


_HOST = {
    "fused": ("ELO_FUSED", lambda v: v != "0", True),                         # fused inference kernels (False: per-operator kernels + GEMMs)
    "cv_prepass": ("ELO_CV_PREPASS", _prepass, None),                         # None: the batch regimes; 0 never, 1 always, N from N rows on
    "select_dense_tiles": ("ELO_SELECT_DENSE_TILES", int, 1024),              # select-k pre-pass: LDS-tiled form from this many 64-centre tiles
    "merge_points": ("ELO_MERGE_POINTS", int, 8192),                          # heterogeneous cost-volume + set-conv launches in forwards of up to this many l0 centres ...
    "merge_level_points": ("ELO_MERGE_LEVEL_POINTS", int, 2000),              # ... at the levels of up to this many centres
    "chain_pair": ("ELO_CHAIN_PAIR", lambda v: v != "0", True),               # cost-volume stage 1 + the level's set-upconv stage 1 as ONE chain-kernel launch where both are chain forms
    "native_submit": ("ELO_NATIVE_SUBMIT", lambda v: v != "0", True),         # a lane submit as ONE C call (copy + hipGraphLaunch on the raw exec handle) instead of torch's copy_ + replay()
    "sv_ride": ("ELO_SV_RIDE", lambda v: v != "0", True),                     # softmax_valid's partial sums ride on the launch that produces its inputs (one launch less per level)
    "train_kernels": ("ELO_TRAIN_KERNELS", lambda v: v != "0", True),
    "train_dense": ("ELO_TRAIN_DENSE", lambda v: v != "0", True),             # a training layer's two products (x W + b with the batch moments, dz W^T) on elo_dense_rows instead of the library GEMM ...
    "train_dense_rows": ("ELO_TRAIN_DENSE_ROWS", int, 25000),                 # ... from this many rows on for x W + b (below: one macro-block per wave, the library + elo_bn_stats are faster: tools/dense_rows_micro.py)
    "train_siamese_batch": ("ELO_TRAIN_SIAMESE_BATCH", lambda v: v != "0", True),   # training: the two frames' feature pyramids as ONE 2B batch whose batch-norm layers keep per-frame statistics (half the encoder's launches)
    "train_dense_fused_dz": ("ELO_TRAIN_DENSE_FUSED_DZ", lambda v: v != "0", True),   # ... and there batch norm's backward-apply happens on that kernel's operand load (dz written on the way): one pass less
    "train_dense_dx_rows": ("ELO_TRAIN_DENSE_DX_ROWS", int, 100000),          # ... and from this many for dz W^T (no moments to fuse: the library holds out longer)
}
# library fields (elo_tuning): name -> (environment variable, default)
_LIB = {
    "chain_forms": ("ELO_CV1_RR", 1), "narrow_mfma": ("ELO_SETCONV_NARROW_MFMA", 1), "range_check": ("ELO_RANGE_CHECK", 0),
    "select_dense_waves": ("ELO_SELECT_DENSE_WAVES", 0), "random_dense_rows": ("ELO_DENSE_ROWS", 0),
    "setconv_chain_rows": ("ELO_SETCONV_RR_ROWS", -1), "mlp_chain_rows": ("ELO_MLP_RR_ROWS", -1), "small_tile_units": ("ELO_SMALL_TILE_UNITS", 64),
    "pool_wave": ("ELO_POOL_WAVE", 1),
}


_LIB_BOOLEAN = ("chain_forms", "narrow_mfma", "range_check", "pool_wave")


def _host_from_env():
    out = {}
    for name, (var, parse, default) in _HOST.items():
        raw = os.environ.get(var)
        out[name] = default if raw is None else parse(raw)
    return out


def lib_from_env():
    """The elo_tuning fields the environment names ({} for the library's defaults): what _lib.lib() installs on load."""
    out = {}
    for name, (var, _default) in _LIB.items():
        raw = os.environ.get(var)
        if raw is not None:
            value = int(raw)
            out[name] = (1 if value else 0) if name in _LIB_BOOLEAN else value      # (ELO_RANGE_CHECK=2 meant "on" to the old atoi != 0)
    return out


_host = _host_from_env()
_version = 0          # bumped by every change made through this module or _lib.set_tuning: model._check_fresh re-hashes only then


def version():
    return _version


def bump():
    global _version
    _version += 1


def get(name):
    return _host[name]


def set_host(name, value):
    """Set a host-side field for the rest of the process (pointnet_util.use_fused); returns the previous value."""
    if name not in _HOST:
        raise KeyError(name)
    prev, _host[name] = _host[name], value
    bump()
    return prev


@contextlib.contextmanager
def override(**fields):
    """`with tuning.override(cv_prepass=1, chain_forms=0):` -- host-side and library fields alike, restored on exit."""
    from . import _lib
    host = {k: v for k, v in fields.items() if k in _HOST}
    lib = {k: v for k, v in fields.items() if k in _LIB}
    unknown = set(fields) - set(host) - set(lib)
    if unknown:
        raise KeyError("unknown tuning field(s): %s" % sorted(unknown))
    prev_host = {k: _host[k] for k in host}
    prev_lib = _lib.get_tuning_base() if lib else None        # (the installed value: a pending elo_debug_* override is not promoted on exit)
    try:
        _host.update(host)
        bump()
        if lib:
            _lib.set_tuning(**lib)
        yield
    finally:
        _host.update(prev_host)
        bump()
        if prev_lib is not None:
            _lib.set_tuning(**prev_lib)


def snapshot():
    """The whole tuning as a plain dict: host-side fields, the library's elo_tuning, the products mode and storage type."""
    from . import _lib, fused
    snap = {k: (dict(v) if isinstance(v, dict) else v) for k, v in _host.items()}
    snap["lib"] = _lib.get_tuning()
    snap["products"] = "half" if fused.products_mode() == fused.PRODUCTS_HALF else "split"
    snap["storage"] = str(fused.storage_dtype()).replace("torch.", "")
    snap["build"] = "fp32-mfma" if fused.fp32_mfma() else "fp16-split"
    return snap


def digest():
    """Hash of the kernel-FORM choices (host-side fields + the library's elo_tuning).  The products mode and the storage type are
    in the snapshot for the record but not in the hash: they are arguments of a capture (a graph keeps the mode it was captured
    under and may be replayed outside the `with` block that set it)."""
    snap = snapshot()
    forms = {k: v for k, v in snap.items() if k not in ("products", "storage", "build")}
    return hashlib.sha256(json.dumps(forms, sort_keys=True).encode()).hexdigest()[:16]
