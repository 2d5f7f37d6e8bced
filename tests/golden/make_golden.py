#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by RUNNING THE REFERENCE (az16/MD_RDM).

Runs only in the build container, where /root/reference exists; the fixtures
(inputs are re-derivable from ``md_rdm_amd.filler``; only expected OUTPUTS are
stored) are committed and travel to the GPU box, the reference does not.

What is executed is the reference's own code:
  * ``network/RDM_Net.py``      DepthEstimationNet / Decoder / Ordinal_Layer / Weights
  * ``network/computations.py`` every post-processing helper
  * ``loss.py``                 Ordinal_Loss
  * ``utils.py``                depth2label_sid
Two things are absent from the checkout and are provided here:
  * torchvision (third-party, unpinned in requirements.txt; era ~0.8-0.9) - the two
    classes the reference calls (``RDM_Net.py:144,526-531``),
    ``torchvision.models.densenet._DenseBlock`` and ``_Transition``, are restated
    below from their published semantics (BN-ReLU-1x1-BN-ReLU-3x3, channel concat;
    BN-ReLU-1x1-AvgPool2), default PyTorch initialisation.
  * ``depth_ratio_008_008_quant.mat`` (.MISSING_LARGE_BLOBS) - derived as 016**2, the
    law the four shipped tables obey.  Only the dormant d_6 path reads
    it; fixtures that depend on it are labelled ``derived008``.
``network/module.py`` needs pytorch_lightning (absent); its 30 lines of step logic
(``module.py:64-97,119-149``) are restated in ``reference_training_step`` on top of
the reference's own cp / utils / loss functions.

Usage:  python tests/golden/make_golden.py            (about 1-2 minutes of CPU)
"""
import os
import shutil
import sys
import tempfile
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

import numpy as np
import scipy.io
import torch
import torch.nn as nn
import torch.nn.functional as F

from md_rdm_amd import filler  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


# --------------------------------------------------------------------------------------
# torchvision.models.densenet stand-in (published semantics of _DenseLayer/_DenseBlock/
# _Transition; memory_efficient only changes *how* backward is computed, not results).
# --------------------------------------------------------------------------------------
class _DenseLayer(nn.Module):
    def __init__(self, num_input_features, growth_rate, bn_size, drop_rate, memory_efficient=False):
        super().__init__()
        self.add_module("norm1", nn.BatchNorm2d(num_input_features))
        self.add_module("relu1", nn.ReLU(inplace=True))
        self.add_module("conv1", nn.Conv2d(num_input_features, bn_size * growth_rate, kernel_size=1, stride=1, bias=False))
        self.add_module("norm2", nn.BatchNorm2d(bn_size * growth_rate))
        self.add_module("relu2", nn.ReLU(inplace=True))
        self.add_module("conv2", nn.Conv2d(bn_size * growth_rate, growth_rate, kernel_size=3, stride=1, padding=1, bias=False))
        self.drop_rate = float(drop_rate)

    def forward(self, inputs):
        x = torch.cat(inputs, 1) if isinstance(inputs, (list, tuple)) else inputs
        y = self.conv1(self.relu1(self.norm1(x)))
        z = self.conv2(self.relu2(self.norm2(y)))
        if self.drop_rate > 0:
            z = F.dropout(z, p=self.drop_rate, training=self.training)
        return z


class _DenseBlock(nn.ModuleDict):
    def __init__(self, num_layers, num_input_features, bn_size, growth_rate, drop_rate, memory_efficient=False):
        super().__init__()
        for i in range(num_layers):
            self.add_module("denselayer%d" % (i + 1),
                            _DenseLayer(num_input_features + i * growth_rate, growth_rate, bn_size, drop_rate, memory_efficient))

    def forward(self, init_features):
        features = [init_features]
        for _, layer in self.items():
            features.append(layer(features))
        return torch.cat(features, 1)


class _Transition(nn.Sequential):
    def __init__(self, num_input_features, num_output_features):
        super().__init__()
        self.add_module("norm", nn.BatchNorm2d(num_input_features))
        self.add_module("relu", nn.ReLU(inplace=True))
        self.add_module("conv", nn.Conv2d(num_input_features, num_output_features, kernel_size=1, stride=1, bias=False))
        self.add_module("pool", nn.AvgPool2d(kernel_size=2, stride=2))


def install_torchvision_standin():
    tv = types.ModuleType("torchvision")
    models = types.ModuleType("torchvision.models")
    densenet = types.ModuleType("torchvision.models.densenet")
    densenet._DenseBlock, densenet._DenseLayer, densenet._Transition = _DenseBlock, _DenseLayer, _Transition
    models.densenet = densenet
    tv.models = models
    sys.modules.update({"torchvision": tv, "torchvision.models": models, "torchvision.models.densenet": densenet})


def prepare_cwd():
    """The reference loads the .mat tables by CWD-relative path (RDM_Net.py:403-407)."""
    d = tempfile.mkdtemp(prefix="rdm_ref_cwd_")
    for s in ("016", "032", "064", "128"):
        shutil.copy(os.path.join(REF, f"depth_ratio_{s}_{s}_quant.mat"), d)
    m16 = scipy.io.loadmat(os.path.join(REF, "depth_ratio_016_016_quant.mat"))
    scipy.io.savemat(os.path.join(d, "depth_ratio_008_008_quant.mat"), {
        "depth_ratio_008_008_quant": m16["depth_ratio_016_016_quant"] ** 2,
        "depth_ratio_008_008_quant_inv": m16["depth_ratio_016_016_quant_inv"] ** 2,
    })
    os.chdir(d)
    return d


def stats3(t):
    t = t.detach().double()
    return np.array([t.mean().item(), t.abs().mean().item(), t.abs().max().item()])


# --------------------------------------------------------------------------------------
# module.py:64-97,119-149 restated over the reference's own functions (is_cuda=False)
# --------------------------------------------------------------------------------------
def reference_training_step(model, x, y, cp, u, l):
    def normalize(batch):                                    # module.py:145-149
        B, C, H, W = batch.size()
        return torch.div(batch, cp.quick_gm(batch.view(B, H * W, 1), H).expand(B, H * W).view(B, 1, H, W))

    y = cp.resize(y, 128)                                    # module.py:68
    gt = y
    mask1 = y > 0
    mask2 = (y <= 0) + 1e-4
    y = (gt * mask1) + mask2                                 # module.py:75-78
    fine_details, ord_depth_pred, ord_label_pred = model(x)  # module.py:80
    yhat_out = [t.detach().clone() for t in fine_details]
    has_ordinal = fine_details[0].shape[2] == 1
    # compute_final_depth, module.py:119-133
    component_target = cp.decompose_depth_map([], normalize(y), 7)[::-1]
    if has_ordinal:
        ord_components = cp.decompose_depth_map([], normalize(u.depth2label_sid(cp.resize(y, 8), cuda=False)), 3)[::-1]
        component_target[0] = ord_components[0]
    target_components = [t.detach().clone() for t in component_target]
    components, fine_detail_loss = cp.optimize_components(fine_details, component_target, False)
    final_depth = cp.recombination(components)
    # compute_ordinal_target, module.py:135-143
    ord_y = u.depth2label_sid(cp.resize(y, ord_depth_pred.shape[2]), cuda=False)
    ord_loss = l.Ordinal_Loss().calc(ord_label_pred, ord_y, cuda=False)
    mse = torch.nn.MSELoss()(final_depth, y)                 # module.py:89
    loss_all = mse + fine_detail_loss + ord_loss             # module.py:90-92
    return dict(loss_all=loss_all, mse=mse, fine_detail_loss=fine_detail_loss, ord_loss=ord_loss,
                final_depth=final_depth, y128=y, ord_y=ord_y, yhat=yhat_out, target_components=target_components,
                ord_depth_pred=ord_depth_pred, ord_label_pred=ord_label_pred)


def run_net_goldens(RDM, cp, u, l, out):
    model = RDM.DepthEstimationNet()
    sd = model.state_dict()
    filler.fill_state_dict(sd)
    keys = list(sd.keys())
    with open(os.path.join(HERE, "state_dict_keys.txt"), "w") as f:
        for k in keys:
            f.write(f"{k} {tuple(sd[k].shape)} {str(sd[k].dtype).replace('torch.', '')}\n")
    n_params = sum(p.numel() for p in model.parameters())
    print("params", n_params, "keys", len(keys))

    taps, tapsub = {}, {}

    def hook(name):
        def fn(mod, inp, outp):
            taps[name] = stats3(outp)
            tapsub[name] = filler.tap_subsample(outp.detach())        # element-wise values on a strided (pixel, channel) lattice
        return fn

    enc = model.encoder
    hs = [enc.conv_e1.register_forward_hook(hook("conv_e1")), enc.max_e1.register_forward_hook(hook("max_e1")),
          enc.dense_e2.register_forward_hook(hook("dense_e2")), enc.trans_e2.register_forward_hook(hook("trans_e2")),
          enc.dense_e3.register_forward_hook(hook("dense_e3")), enc.trans_e3.register_forward_hook(hook("trans_e3")),
          enc.dense_e4.register_forward_hook(hook("dense_e4")), enc.trans_e4.register_forward_hook(hook("trans_e4")),
          model.d_1.dense_layer.register_forward_hook(hook("d1_dense")), model.d_1.conv2.register_forward_hook(hook("d1_conv2"))]
    logits = {}
    hs.append(model.d_1.conv2.register_forward_hook(lambda m, i, o: logits.__setitem__("v", o.detach().clone())))

    # ---- (1) train-mode full step, B=2, 228x228 -------------------------------------
    B, H, W = 2, 228, 228
    xn, yn = filler.synthetic_batch(B, H, W, seed=filler.MARGIN_SEEDS["train228"])
    x, y = torch.from_numpy(xn), torch.from_numpy(yn)
    model.train()
    r = reference_training_step(model, x, y, cp, u, l)
    r["loss_all"].backward()
    g = {}
    g["train228_decode_c"] = r["ord_depth_pred"].numpy()
    g["train228_ord_labels"] = r["ord_label_pred"].detach().numpy()
    g["train228_logits"] = logits["v"].numpy()
    assert filler.dorn_unsafe_pairs(g["train228_logits"]) == 0, "train228 seed has near-tie ordinal pairs: pick another (filler.MARGIN_SEEDS)"
    for i, t in enumerate(r["yhat"]):
        g[f"train228_yhat{i}"] = t.numpy()
    for i, t in enumerate(r["target_components"]):
        g[f"train228_target_comp{i}_stats"] = stats3(t)
    g["train228_target_comp0"] = r["target_components"][0].numpy()
    g["train228_target_comp3"] = r["target_components"][3].numpy()
    g["train228_final_depth_stats"] = stats3(r["final_depth"])
    g["train228_final_depth_corner"] = r["final_depth"].detach()[:, :, :4, :4].numpy()
    g["train228_ord_y"] = r["ord_y"].numpy()
    g["train228_losses"] = np.array([r["mse"].item(), float(r["fine_detail_loss"]), r["ord_loss"].item(), r["loss_all"].item()])
    for k, v in taps.items():
        g[f"train228_tap_{k}"] = v
        g[f"train228_tapsub_{k}"] = tapsub[k]
    names, gnorm, gsum, ghead = [], [], [], []
    for name, p in model.named_parameters():
        names.append(name)
        if p.grad is None:
            gnorm.append(-1.0); gsum.append(0.0); ghead.append(np.zeros(4))
        else:
            gd = p.grad.detach().double().flatten()
            gnorm.append(gd.norm().item()); gsum.append(gd.sum().item())
            hh = np.zeros(4); hh[:min(4, gd.numel())] = gd[:4].numpy(); ghead.append(hh)
    g["train228_grad_names"] = np.array(names)
    g["train228_grad_norm"] = np.array(gnorm)
    g["train228_grad_sum"] = np.array(gsum)
    g["train228_grad_head"] = np.stack(ghead)
    # full gradients of a few small tensors
    for name in ["encoder.conv_e1.weight", "encoder.conv_e1.bias", "d_1.conv2.bias", "encoder.dense_e2.denselayer1.norm1.weight",
                 "encoder.dense_e2.denselayer1.norm1.bias", "encoder.trans_e3.norm.weight", "d_1.dense_layer.denselayer24.norm2.bias",
                 "weight_layer.d0", "weight_layer.f1", "weight_layer.f2", "weight_layer.f3"]:
        g["train228_grad__" + name] = dict(model.named_parameters())[name].grad.numpy()
    # BN running statistics after ONE train forward (momentum 0.1, unbiased var)
    sd2 = model.state_dict()
    for name in ["encoder.dense_e2.denselayer1.norm1", "encoder.dense_e2.denselayer6.norm2", "encoder.trans_e2.norm",
                 "encoder.dense_e4.denselayer36.norm1", "d_1.dense_layer.denselayer24.norm2"]:
        g["train228_rm__" + name] = sd2[name + ".running_mean"].numpy().copy()
        g["train228_rv__" + name] = sd2[name + ".running_var"].numpy().copy()
        g["train228_nbt__" + name] = sd2[name + ".num_batches_tracked"].numpy().copy()
    model.zero_grad()

    # ---- (2) eval-mode forward, B=1, 226x226 (the size module.py:19,24 feeds) -------
    filler.fill_state_dict(model.state_dict())
    model.eval()
    xn, _ = filler.synthetic_batch(1, 226, 226, seed=filler.MARGIN_SEEDS["eval226"])
    with torch.no_grad():
        yh, dc, ol = model(torch.from_numpy(xn))
    g["eval226_decode_c"] = dc.numpy(); g["eval226_ord_labels"] = ol.numpy(); g["eval226_logits"] = logits["v"].numpy()
    assert filler.dorn_unsafe_pairs(g["eval226_logits"]) == 0, "eval226 seed has near-tie ordinal pairs"
    for i, t in enumerate(yh):
        g[f"eval226_yhat{i}"] = t.numpy()
    for k, v in taps.items():
        g[f"eval226_tap_{k}"] = v
        g[f"eval226_tapsub_{k}"] = tapsub[k]

    # ---- (3) rectangular geometries: reference is defined up to the DORN head only ---
    #      (RDM_Net.py:73-103; decompose raises afterwards)
    def head_only(xt):
        e = model.encoder
        t = e.conv_e1(xt); t = e.max_e1(t); t = e.dense_e2(t); t = e.pad_br(t); t = e.trans_e2(t)
        t = e.dense_e3(t); t = e.pad_br(t); t = e.trans_e3(t); t = e.dense_e4(t); t = e.pad_br(t); t = e.trans_e4(t)
        return model.d_1(t)

    model.train()
    filler.fill_state_dict(model.state_dict())
    xn, _ = filler.synthetic_batch(2, 228, 304, seed=filler.MARGIN_SEEDS["train228x304"])
    with torch.no_grad():
        dc, ol = head_only(torch.from_numpy(xn))
    g["train228x304_decode_c"] = dc.numpy(); g["train228x304_ord_labels"] = ol.numpy(); g["train228x304_logits"] = logits["v"].numpy()
    assert filler.dorn_unsafe_pairs(g["train228x304_logits"]) == 0, "train228x304 seed has near-tie ordinal pairs"
    for k, v in taps.items():
        g[f"train228x304_tap_{k}"] = v
        g[f"train228x304_tapsub_{k}"] = tapsub[k]
    try:
        model(torch.from_numpy(xn))
        g["train228x304_full_forward_raises"] = np.array(0)
    except RuntimeError as e:
        g["train228x304_full_forward_raises"] = np.array(1)
        print("228x304 full forward raises as expected:", str(e)[:80])
    for h in hs:
        h.remove()
    g["seeds"] = np.array([filler.MARGIN_SEEDS[k] for k in ("train228", "eval226", "train228x304")])
    np.savez_compressed(os.path.join(out, "net_goldens.npz"), **g)
    print("net goldens:", len(g), "arrays")
    return model


def run_wsm_goldens(RDM, out):
    """Dormant WSM decoder blocks (RDM_Net.py:163-236) as standalone operators, reduced width."""
    g = {}
    torch.manual_seed(0)
    for (cin, k, lid, raw, hw) in [(64, 16, 2, 128, 8), (32, 32, 3, 64, 16)]:
        m = RDM.WSMLayer(cin, k, k, lid)
        sd = m.state_dict()
        for key, t in sd.items():
            t.copy_(torch.from_numpy(filler.state_value(f"wsm{lid}." + key, tuple(t.shape))))
        x = torch.from_numpy(filler.uniform(f"wsm{lid}.x", (2, raw, hw, hw), -1.0, 1.0))
        x.requires_grad_(True)
        y = m(x)
        (y * torch.from_numpy(filler.uniform(f"wsm{lid}.gy", tuple(y.shape), -1.0, 1.0))).sum().backward()
        g[f"wsm{lid}_out"] = y.detach().numpy()
        g[f"wsm{lid}_dx_stats"] = stats3(x.grad)
        g[f"wsm{lid}_dx_head"] = x.grad[0, :4, :2, :2].numpy()
        for name, p in m.named_parameters():
            g[f"wsm{lid}_gradnorm__{name}"] = np.array(p.grad.double().norm().item())
        g[f"wsm{lid}_keys"] = np.array([f"{k_} {tuple(v.shape)}" for k_, v in sd.items()])
    np.savez_compressed(os.path.join(out, "wsm_goldens.npz"), **g)
    print("wsm goldens:", len(g))


def run_op_goldens(RDM, cp, u, l, out):
    g = {}
    U = filler.uniform
    LU = filler.log_uniform
    tt = torch.from_numpy

    # quick_gm (computations.py:244-255) - squares its rc argument
    d = LU("op.gm", (3, 64, 1), 0.5, 2.0)
    g["quick_gm_8"] = cp.quick_gm(tt(d), 8).numpy()
    di = np.floor(U("op.gmi", (3, 64, 1), 1, 60)).astype(np.int64)
    g["quick_gm_int"] = cp.quick_gm(tt(di), 8).numpy()
    # resize (bicubic f64, :308-311) for every size pair the path uses + rectangles
    for (h, w, s) in [(8, 8, 4), (4, 4, 2), (2, 2, 1), (8, 8, 8), (226, 226, 128), (128, 128, 64), (16, 16, 8), (8, 10, 8), (8, 10, 4),
                      (228, 304, 128), (128, 128, 8), (32, 32, 16), (11, 38, 8)]:
        src = LU(f"op.rs{h}x{w}", (2, 1, h, w), 0.5, 9.5)
        g[f"resize_{h}x{w}_to_{s}"] = cp.resize(tt(src), s).numpy()
    # upsample / multi_upsample (:357-366)
    src = U("op.up", (2, 1, 4, 4), 0.5, 2.0)
    g["multi_upsample_4_n3"] = cp.multi_upsample(tt(src), 3).numpy()
    # decompose_depth_map (:368-392) n=3 and n=7
    src = LU("op.dec8", (2, 1, 8, 8), 0.5, 2.0).astype(np.float64)
    for i, t in enumerate(cp.decompose_depth_map([], tt(src), 3)[::-1]):
        g[f"decompose3_{i}"] = t.numpy()
    src = LU("op.dec128", (2, 1, 128, 128), 0.5, 9.5)
    comps = cp.decompose_depth_map([], tt(src), 7)[::-1]
    for i, t in enumerate(comps):
        g[f"decompose7_{i}_stats"] = stats3(t)
    g["decompose7_0"] = comps[0].numpy(); g["decompose7_3"] = comps[3].numpy(); g["decompose7_7_corner"] = comps[7][:, :, :6, :6].numpy()
    src_rel = LU("op.decrel", (2, 1, 16, 16), 0.5, 2.0).astype(np.float64)
    rel = cp.decompose_depth_map([], tt(src_rel), 4, relative_map=True)[::-1]
    g["decompose4_rel_len"] = np.array(len(rel)); g["decompose4_rel_0"] = rel[0].numpy()
    # relative_fine_detail_matrix + make_pred (:423-484, :512-528) with one and with two candidates per level
    f1 = [tt(LU(f"op.fd1_{i}", (2, 1, 2 ** i, 2 ** i), 0.5, 2.0).astype(np.float64)) for i in range(4)]
    f2 = [tt(LU(f"op.fd2_{i}", (2, 1, 2 ** i, 2 ** i), 0.5, 2.0).astype(np.float64)) for i in range(1, 4)]
    mats = cp.relative_fine_detail_matrix([f1, f2], False)
    for i, m in enumerate(mats):
        g[f"rfdm_{i}"] = m.numpy()
    w = [tt(U("op.w0", (1, 1), 0.5, 1.5)), tt(U("op.w1", (2, 1), 0.2, 0.8)), tt(U("op.w2", (2, 1), 0.2, 0.8)), tt(U("op.w3", (2, 1), 0.2, 0.8))]
    pred = cp.make_pred(w, [m.clone() for m in mats], False, False)
    for i, p in enumerate(pred):
        g[f"make_pred_{i}"] = p.numpy()
    # recombination (:394-421) with d_0 and without
    comps = [tt(U(f"op.rc{i}", (2, 1, 2 ** i, 2 ** i), -1.0, 1.0).astype(np.float64)) for i in range(4)]
    g["recombination_n7"] = cp.recombination([c.clone() for c in comps], 7)[:, :, ::16, ::16].numpy()
    g["recombination_n3"] = cp.recombination([c.clone() for c in comps], 3).numpy()
    g["recombination_n4_rel"] = cp.recombination([c.clone() for c in comps[1:]], 4).numpy()
    # optimize_components / squared_err (:499-510,:530-544)
    yh = [tt(U(f"op.oc_yh{i}", (2, 1, 2 ** i, 2 ** i), -1, 1)) for i in range(4)]
    yt = [tt(U(f"op.oc_y{i}", (2, 1, 2 ** i, 2 ** i), -1, 1).astype(np.float64)) for i in range(8)]
    _, lsum = cp.optimize_components(yh, yt, False)
    g["optimize_components_loss"] = np.array(float(lsum))
    # depth2label_sid (utils.py:195-211) incl. values below alpha
    dep = np.concatenate([LU("op.sid", (60,), 0.005, 12.0), np.array([1e-4, 0.02, 10.0, 0.0199999], dtype=np.float32)]).reshape(1, 1, 8, 8).astype(np.float64)
    g["depth2label_sid"] = u.depth2label_sid(tt(dep)).numpy()
    g["depth2label_sid_f32"] = u.depth2label_sid(tt(dep.astype(np.float32))).numpy()
    # Ordinal_Loss (loss.py:8-59)
    P = U("op.ol_p", (2, 90, 8, 8), 0.0, 1.0).astype(np.float64)
    P.flat[::97] = 0.0; P.flat[5::101] = 1.0
    T = np.floor(U("op.ol_t", (2, 1, 8, 8), 0, 95)).astype(np.int32)
    Pt = tt(P).requires_grad_(True)
    lo = l.Ordinal_Loss().calc(Pt, tt(T), cuda=False)
    lo.backward()
    g["ordinal_loss"] = np.array(lo.item()); g["ordinal_loss_dP"] = Pt.grad.numpy()
    # DornOrdinalRegression (RDM_Net.py:313-345) incl. clamp edges
    ol_ = RDM.Ordinal_Layer(1, True, None)
    xl = U("op.dorn", (2, 180, 8, 10), -2.0, 3.0)
    xl.flat[::53] = 2e4; xl.flat[7::59] = -5.0; xl[0, 10, 0, 0] = xl[0, 11, 0, 0]
    xlt = tt(xl).requires_grad_(True)
    dec, lab = ol_.DornOrdinalRegression(xlt)
    (lab * tt(U("op.dorn_g", (2, 90, 8, 10), -1, 1).astype(np.float64))).sum().backward()
    g["dorn_decode"] = dec.numpy(); g["dorn_labels"] = lab.detach().numpy(); g["dorn_dx"] = xlt.grad.numpy()

    # ---- dormant relative decoders ---------------------------------------------------
    quant = RDM.Quantization()
    # sparse_comparison_v1 (RDM_Net.py:244-257) + Lloyd id=3 [derived008]
    o6 = RDM.Ordinal_Layer(6, False, quant)
    d3 = LU("op.d3", (2, 1, 8, 8), 0.5, 2.0)
    g["derived008_sparse_v1"] = o6.sparse_comparison_v1(tt(d3)).numpy()
    g["derived008_quadratic_als"] = cp.quadratic_als(tt(g["derived008_sparse_v1"]), cuda=False, n=3).numpy()
    g["derived008_d6_forward"] = o6(tt(d3)).numpy()
    # sparse_comparison_id (RDM_Net.py:259-284) + Lloyd (016 table), id=4
    o7 = RDM.Ordinal_Layer(7, False, quant)
    dn = LU("op.dn16", (2, 1, 16, 16), 0.5, 2.0)
    dn1 = cp.resize(tt(dn), 8)
    g["resize_dn16_to_8"] = dn1.numpy()
    sp = o7.sparse_comparison_id(tt(dn), dn1)
    g["sparse_id_016"] = sp.numpy(); g["sparse_id_016_dtype"] = np.array(str(sp.dtype))
    # raw (un-quantised) ratio grid, obtained from the reference's own get_resized_area
    raw = []
    for r_ in range(16):
        for c_ in range(16):
            rs = int(min(max(np.floor(r_ / 2), 0), 8 - 3)); cs = int(min(max(np.floor(c_ / 2), 0), 8 - 3))
            area = cp.get_resized_area(rs, rs + 2, cs, cs + 3, dn1)
            raw.append(tt(dn).view(2, 16, 16)[:, r_, c_].view(2, 1, 1) * torch.pow(area, -1))
    g["ratio_grid_raw_16"] = torch.cat(raw, 1).numpy()
    als = cp.alternating_least_squares(sparse_m=sp, n=4, limit=100, cuda=False)
    g["als_016"] = als.numpy()
    g["d7_forward"] = o7(tt(dn)).numpy()
    # ALS on a generic positive matrix, several limits (first-min selection, batch-global rmse)
    R = LU("op.alsR", (3, 256, 64), 0.5, 2.0)
    for lim in (1, 5, 30, 100):
        g[f"als_generic_limit{lim}"] = cp.alternating_least_squares(tt(R), n=4, limit=lim, cuda=False).numpy()
    R8 = LU("op.alsR8", (2, 64, 64), 0.5, 2.0)
    g["quadratic_als_generic"] = cp.quadratic_als(tt(R8), cuda=False, n=3).numpy()
    # als_step alone (:175-193)
    q = tt(LU("op.alsq", (3, 64, 1), 0.5, 2.0))
    g["als_step_p"] = cp.als_step(tt(R), q, cuda=False).numpy()
    # Lloyd for every shipped table on a raw grid that spans the whole threshold range
    for oid, s in [(7, "016"), (8, "032"), (9, "064"), (10, "128")]:
        o = RDM.Ordinal_Layer(oid, False, quant)
        rr = LU(f"op.lloyd{s}", (1, 32, 16), 0.2, 5.0).astype(np.float64)
        qv, inv = quant.get_with_id(o.id)
        rr.flat[:40] = qv[:, 0]                       # exactly on every threshold (>= is inclusive)
        rr.flat[40:80] = np.nextafter(qv[:, 0], 0)    # one ulp below
        lab = torch.zeros(1, 32, 16, 40)
        g[f"lloyd_{s}"] = o.LloydQuantization(lab, tt(rr).clone(), id=o.id).numpy()
    # split_matrix / reconstruct (:201-238) + the d_8 path (32x32, 4 pages)
    d32 = LU("op.d32", (2, 1, 32, 32), 0.5, 2.0)
    d16 = cp.resize(tt(d32), 16)
    a, b = cp.split_matrix(tt(d32), d16)
    g["split_len"] = np.array(len(a)); g["split_first_2"] = a[2].numpy(); g["split_second_3"] = b[3].numpy()
    g["reconstruct_4pages"] = cp.reconstruct([tt(U(f"op.pg{i}", (2, 1, 16, 16), 0, 1)) for i in range(4)]).numpy()
    o8 = RDM.Ordinal_Layer(8, False, quant)
    g["d8_forward"] = o8(tt(d32)).numpy()
    np.savez_compressed(os.path.join(out, "op_goldens.npz"), **g)
    print("op goldens:", len(g), "arrays")



def run_rel_goldens(RDM, cp, out):
    """the relative decoders d_6..d_9 as the reference's own classes build them (RDM_Net.py:57-60 are commented
    out, but ``Decoder(in_channels=1056, num_wsm_layers=k, DORN=False, id=6+k, ...)`` is live code), at FULL width, fed a
    deterministic encoder output, and the five-decoder tail of forward() (:106-133) composed from the reference's functions.
    conv1's bias is set to +2 with small weights so the 1-channel map is positive (nothing in the reference constrains it;
    ratios, logs and geometric means of a non-positive map are NaN)."""
    g = {}
    tt = torch.from_numpy
    quant = RDM.Quantization()
    B = 2
    x = tt(filler.uniform("rel.x", (B, 1056, 8, 8), -1.0, 1.0))
    rows = []
    cnt = np.floor(filler.uniform("rel.cnt", (B, 1, 8, 8), 5, 80)).astype(np.int64)          # a DORN count map for d_1
    x_d1 = tt(cnt)
    f_d1 = cp.decompose_depth_map([], torch.div(x_d1, cp.quick_gm(x_d1.view(B, 64, 1), 8).expand(B, 64).view(B, 1, 8, 8)), 3)[::-1]
    rows.append(f_d1)
    for did in (6, 7, 8, 9):
        torch.manual_seed(did)
        dec = RDM.Decoder(in_channels=1056, num_wsm_layers=did - 6, DORN=False, id=did, quant=quant)
        sd = dec.state_dict()
        for key, t in sd.items():
            if t.numel() and t.dtype.is_floating_point:
                t.copy_(tt(filler.state_value(f"d_{did}." + key, tuple(t.shape))))
        with torch.no_grad():
            dec.conv1.weight.mul_(0.02)
            dec.conv1.bias.fill_(2.0)
        dec.train()
        with torch.no_grad():
            h = dec.dense_layer(x)
            g[f"rel{did}_dense_stats"] = stats3(h)
            g[f"rel{did}_dense_head"] = h[:, -48:, :2, :2].numpy()
            h = dec.wsm_block(h)
            g[f"rel{did}_wsm_stats"] = stats3(h)
            feat = dec.conv1(h)
            g[f"rel{did}_feat"] = feat.numpy()
            outm = dec.ord_layer(feat.clone())
        g[f"rel{did}_out"] = outm.numpy()
        g[f"rel{did}_rm_norm2_24"] = dec.dense_layer.denselayer24.norm2.running_mean.numpy().copy()
        g[f"rel{did}_rv_norm1_1"] = dec.dense_layer.denselayer1.norm1.running_var.numpy().copy()
        g[f"rel{did}_keys"] = np.array([f"{k_} {tuple(v.shape)}" for k_, v in sd.items()])
        n = int(np.log2(outm.shape[2]))
        rows.append(cp.decompose_depth_map([], outm, n, relative_map=True)[::-1])
        print(f"  d_{did}: feat {tuple(feat.shape)} out {tuple(outm.shape)} range [{float(outm.min()):.3f}, {float(outm.max()):.3f}]", flush=True)
    # ---- d_10 (RDM_Net.py:61: four WSM layers, 208 channels at 128x128, 64 pages of 16x16) on its own: features and head -------
    did = 10
    torch.manual_seed(did)
    dec = RDM.Decoder(in_channels=1056, num_wsm_layers=4, DORN=False, id=did, quant=quant)
    sd = dec.state_dict()
    for key, t in sd.items():
        if t.numel() and t.dtype.is_floating_point:
            t.copy_(tt(filler.state_value(f"d_{did}." + key, tuple(t.shape))))
    with torch.no_grad():
        dec.conv1.weight.mul_(0.02)
        dec.conv1.bias.fill_(2.0)
    dec.train()
    with torch.no_grad():
        h = dec.wsm_block(dec.dense_layer(x))
        g["rel10_wsm_stats"] = stats3(h)
        feat = dec.conv1(h)
        g["rel10_feat_stats"] = stats3(feat)
        g["rel10_feat_sub"] = feat[:, :, ::8, ::8].numpy()
        outm = dec.ord_layer(feat.clone())
    g["rel10_out"] = outm.numpy()
    g["rel10_feat"] = feat.numpy().astype(np.float32)
    print(f"  d_10: feat {tuple(feat.shape)} out {tuple(outm.shape)} range [{float(outm.min()):.3f}, {float(outm.max()):.3f}]", flush=True)
    g["rel_row_lens"] = np.array([len(r) for r in rows])
    y_hat = cp.relative_fine_detail_matrix(rows, False)
    g["rel_matrix_shapes"] = np.array([list(m.shape) for m in y_hat])
    sizes = [m.shape[1] for m in y_hat] + [0] * (8 - len(y_hat))
    g["rel_vector_sizes"] = np.array(sizes)
    torch.manual_seed(0)
    wl = RDM.Weights(vector_sizes=sizes, use_cuda=False, relative_only=False)
    with torch.no_grad():
        for i, w in enumerate(wl.weight_list):
            if w.numel():
                w.copy_(tt(filler.uniform(f"rel.w{i}", tuple(w.shape), 0.1, 0.6)))
    pred = wl(y_hat)
    for i, t in enumerate(pred):
        g[f"rel_yhat{i}"] = t.detach().numpy()
    (sum((t.double() ** 2).sum() for t in pred)).backward()
    for i, w in enumerate(wl.weight_list):
        if w.numel():
            g[f"rel_dw{i}"] = w.grad.numpy()
    comps = [t.detach() for t in pred]
    g["rel_recombination"] = cp.recombination(comps, n=7).numpy()
    np.savez_compressed(os.path.join(out, "rel_goldens.npz"), **g)
    print("rel goldens:", len(g), "arrays")


def install_lightning_standin():
    """LABELLED STAND-IN, not the reference and not pytorch_lightning: ``metrics.py`` does ``import pytorch_lightning as pl``,
    ``from pytorch_lightning.metrics.metric import Metric`` and reads ``pl.metrics.functional.__dict__`` for three library
    metrics ('mean_squared_error', 'mean_squared_log_error', 'mean_absolute_error', metrics.py:118-121; pinned 1.1.7 in
    requirements.txt:1, absent from this image).  Those three are restated from their published definitions; everything else that
    produces the fixture - MetricComputation.compute (clamp 1e-7, target > 0 mask, :58-66) and the seven metric functions the
    reference defines itself (:79-116) - is the reference's own code, imported and run."""
    pl = types.ModuleType("pytorch_lightning")
    metrics = types.ModuleType("pytorch_lightning.metrics")
    metric = types.ModuleType("pytorch_lightning.metrics.metric")
    functional = types.ModuleType("pytorch_lightning.metrics.functional")

    class Metric(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
    metric.Metric = Metric
    functional.mean_squared_error = lambda pred, target: torch.mean((pred - target) ** 2)
    functional.mean_absolute_error = lambda pred, target: torch.mean(torch.abs(pred - target))
    functional.mean_squared_log_error = lambda pred, target: torch.mean((torch.log1p(pred) - torch.log1p(target)) ** 2)
    metrics.metric, metrics.functional = metric, functional
    pl.metrics = metrics
    sys.modules.update({"pytorch_lightning": pl, "pytorch_lightning.metrics": metrics, "pytorch_lightning.metrics.metric": metric,
                        "pytorch_lightning.metrics.functional": functional})


def run_metric_goldens(out):
    """the validation metrics (metrics.py:48-128) on deterministic maps - invalid (zero) target pixels, predictions
    below the 1e-7 clamp, float64 (the dtype validation_step hands over, module.py:99-117) and float32."""
    install_lightning_standin()
    import metrics as ref_metrics
    names = ["delta1", "delta2", "delta3", "mse", "mae", "log10", "absrel", "sqrel", "rmse"]
    g = {"names": np.array(names)}
    for tag, shape in [("a", (4, 1, 128, 128)), ("b", (1, 1, 8, 8))]:
        pred = filler.uniform(f"met.p.{tag}", shape, -0.5, 3.0).astype(np.float64)
        tgt = filler.log_uniform(f"met.t.{tag}", shape, 0.2, 4.0).astype(np.float64)
        tgt.flat[::7] = 0.0
        for dt in (np.float64, np.float32):
            mc = ref_metrics.MetricComputation(names)
            vals = mc.compute(torch.from_numpy(pred.astype(dt)), torch.from_numpy(tgt.astype(dt)))
            g[f"metrics_{tag}_{np.dtype(dt).name}"] = np.array([float(v) for v in vals], dtype=np.float64)
            g[f"metrics_{tag}_{np.dtype(dt).name}_avg_delta1"] = np.array(float(mc.avg("delta1")))
    np.savez_compressed(os.path.join(out, "metric_goldens.npz"), **g)
    print("metric goldens:", len(g), "arrays")


def main():
    install_torchvision_standin()
    cwd = prepare_cwd()
    sys.path.insert(0, REF)
    import network.RDM_Net as RDM
    import network.computations as cp
    import utils as u
    import loss as l
    RDM.use_cuda = False            # global read at call time (RDM_Net.py:63,100)
    out = HERE
    which = sys.argv[1:] or ["ops", "wsm", "rel", "net", "metrics"]
    if "ops" in which:
        run_op_goldens(RDM, cp, u, l, out)
    if "wsm" in which:
        run_wsm_goldens(RDM, out)
    if "rel" in which:
        run_rel_goldens(RDM, cp, out)
    if "net" in which:
        run_net_goldens(RDM, cp, u, l, out)
    if "metrics" in which:
        run_metric_goldens(out)
    shutil.rmtree(cwd, ignore_errors=True)


if __name__ == "__main__":
    main()
