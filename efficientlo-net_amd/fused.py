"""Host side of the fused inference kernels (csrc/elo_fused.hip, include/elo.h "Fused inference
kernels"): weight packing into MFMA B-fragment order and torch-tensor front-ends.

A packed layer is cached in the VariableStore next to the folded weights and dropped by
VariableStore.invalidate(); packing costs a few small torch ops once per parameter update."""
import torch

from . import _lib as L
from . import tf_util


class PackedDense:
    """One inference layer y = act(x @ W + b) with BN folded, packed for v_mfma_f32_16x16x4_f32."""

    def __init__(self, W, b, relu=True, row_order=None):
        if row_order is not None:                         # the kernel's LDS column order differs from the concat order
            W = W[row_order]
        K, N = W.shape
        Kp, Np = (K + 15) // 16 * 16, (N + 15) // 16 * 16
        Wp = torch.zeros((Kp, Np), dtype=torch.float32, device=W.device)
        Wp[:K, :N] = W
        # packed[((cb*KS + ks)*64 + lane)*4 + s] = Wp[ks*16 + 4*(lane>>4) + s][cb*16 + (lane&15)]
        self.w = Wp.reshape(Kp // 16, 4, 4, Np // 16, 16).permute(3, 0, 1, 4, 2).contiguous()
        self.b = torch.zeros((Np,), dtype=torch.float32, device=W.device)
        self.b[:N] = b
        self.K, self.N, self.relu = K, N, relu

    def struct(self):
        return L.Dense(self.w.data_ptr(), self.b.data_ptr(), self.K, self.N, 1 if self.relu else 0)


def packed_layer(scope, cin, cout, bn=True, relu=True, row_order=None, tf_kernel_dims=(1, 1)):
    """get-or-create the layer's variables under the active scope, fold, pack, cache."""
    store = tf_util.get_store()
    name, W, b, bn_vars = tf_util.dense_variables(scope, cin, cout, tf_kernel_dims, bn)
    key = ("packed", name, None if row_order is None else tuple(row_order), relu)
    hit = store._folded.get(key)
    if hit is None:
        Wf, bf = store.folded(name, W, b, bn_vars)
        order = None if row_order is None else torch.as_tensor(row_order, device=Wf.device)
        hit = PackedDense(Wf, bf, relu, order)
        store._folded[key] = hit
    return hit


def _f32c(t):
    if t.dtype != torch.float32:
        raise TypeError("feature-path tensors are float32")
    return t.contiguous()


def _chain(layers):
    arr = (L.Dense * 3)()
    for i, p in enumerate(layers):
        arr[i] = p.struct()
    return arr


def setconv(src_xyz, src_feat, idx, mask, layers, centre_xyz=None, xyz1_grid=None, centre_hw=None):
    """group_concat -> MLP chain -> masked max over K in one launch.
    Returns (out (B,n,Cout), new_xyz (B,n,3) or None)."""
    L.require_gpu(src_xyz, src_feat, idx, mask, centre_xyz, xyz1_grid, centre_hw)
    src_xyz, src_feat, mask = _f32c(src_xyz), _f32c(src_feat), _f32c(mask)
    idx = idx.contiguous()
    B, n, K, _ = idx.shape
    _, H2, W2, C = src_feat.shape
    dev = idx.device
    out = torch.empty((B, n, layers[-1].N), dtype=torch.float32, device=dev)
    ptr = lambda x: x.data_ptr() if x is not None else None
    if centre_hw is not None:
        xyz1_grid, centre_hw = _f32c(xyz1_grid), centre_hw.contiguous()
        H, W = xyz1_grid.shape[1:3]
        new_xyz = torch.empty((B, n, 3), dtype=torch.float32, device=dev)
    else:
        centre_xyz = _f32c(centre_xyz)
        H = W = 0
        new_xyz = None
    a = L.SetconvArgs(B, n, K, H, W, H2, W2, C, ptr(xyz1_grid), ptr(centre_hw), ptr(centre_xyz), src_xyz.data_ptr(),
                      src_feat.data_ptr(), idx.data_ptr(), mask.data_ptr(), len(layers), _chain(layers),
                      out.data_ptr(), ptr(new_xyz))
    L.call("elo_setconv_fused", a, out)
    return out, new_xyz


def mlp(sources, layers):
    """Row-wise MLP over concat(sources, -1) without building the concat.  sources: (..., C_i) tensors."""
    L.require_gpu(*sources)
    lead = sources[0].shape[:-1]
    srcs = [_f32c(s).reshape(-1, s.shape[-1]) for s in sources]
    rows = srcs[0].shape[0]
    out = torch.empty((rows, layers[-1].N), dtype=torch.float32, device=srcs[0].device)
    a = L.MlpArgs()
    a.rows, a.n_sources, a.n_layers, a.layers, a.out = rows, len(srcs), len(layers), _chain(layers), out.data_ptr()
    for i, s in enumerate(srcs):
        a.src[i], a.src_width[i] = s.data_ptr(), s.shape[1]
    L.call("elo_mlp_fused", a, out)
    return out.reshape(lead + (layers[-1].N,))


def cv_stage1(xyz1, feat1, xyz2_proj, feat2_proj, idx, mask, cv0, cv1, cv2, cv_xyz, sum_cv0, sum_cv1):
    L.require_gpu(xyz1, feat1, xyz2_proj, feat2_proj, idx, mask)
    xyz1, feat1, xyz2_proj, feat2_proj, mask = (_f32c(t) for t in (xyz1, feat1, xyz2_proj, feat2_proj, mask))
    idx = idx.contiguous()
    B, N, K, _ = idx.shape
    _, H2, W2, C = feat2_proj.shape
    out = torch.empty((B, N, 64), dtype=torch.float32, device=idx.device)
    a = L.Cv1Args(B, N, K, H2, W2, C, xyz1.data_ptr(), feat1.data_ptr(), xyz2_proj.data_ptr(), feat2_proj.data_ptr(),
                  idx.data_ptr(), mask.data_ptr(), cv0.struct(), cv1.struct(), cv2.struct(), cv_xyz.struct(),
                  sum_cv0.struct(), sum_cv1.struct(), out.data_ptr())
    L.call("elo_cv_stage1_fused", a, out)
    return out


def cv_stage2(xyz1_proj, feat1_proj, cost_proj, idx, mask, xyz_enc, sum_cost0, sum_cost1):
    L.require_gpu(xyz1_proj, feat1_proj, cost_proj, idx, mask)
    xyz1_proj, feat1_proj, cost_proj, mask = (_f32c(t) for t in (xyz1_proj, feat1_proj, cost_proj, mask))
    idx = idx.contiguous()
    B, N, K, _ = idx.shape
    _, H, W, C = feat1_proj.shape
    out = torch.empty((B, N, 64), dtype=torch.float32, device=idx.device)
    a = L.Cv2Args(B, N, K, H, W, C, xyz1_proj.data_ptr(), feat1_proj.data_ptr(), cost_proj.data_ptr(), idx.data_ptr(),
                  mask.data_ptr(), xyz_enc.struct(), sum_cost0.struct(), sum_cost1.struct(), out.data_ptr())
    L.call("elo_cv_stage2_fused", a, out)
    return out
