"""ORACLE (test infrastructure, never shipped or measured as product).

CPU restatement, in plain PyTorch-fp32 functional ops, of the reference's AdaPose stereo
pose network `StereoPoseNet_with_depth.forward`
(`/root/reference/models/pose_estimator/AdaPose/lib/network_v5.py:418-519`) in eval mode
(BatchNorm3d uses running stats, Dropout2d is identity).

Parity pin: `tests/test_oracle_golden.py` checks this file against `tests/golden/adapose_b2.npz`,
which `tools/make_goldens.py` produced by importing the reference module itself in the build
container (the reference has no tests / golden vectors of its own).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
PSP_BINS = (1, 2, 3, 6)
# (planes, blocks, stride of block 0, dilation of blocks >= 1)   pspnet.py:40-43,53-63
RESNET34_LAYERS = ((64, 3, 1, 1), (128, 4, 2, 1), (256, 6, 1, 2), (512, 3, 1, 4))


def to_torch_sd(sd, prefix_strip="module."):
    out = {}
    for k, v in sd.items():
        if k.startswith(prefix_strip):
            k = k[len(prefix_strip):]
        out[k] = v if isinstance(v, torch.Tensor) else torch.from_numpy(v)
    return out


# ------------------------------------------------------------------ PSPNet (pspnet.py)
def basic_block(x, sd, p, stride, dil):
    """pspnet.py:21-30 — conv/relu/conv (+1x1 downsample) + residual, relu; no BN."""
    out = F.relu(F.conv2d(x, sd[p + "conv1.weight"], None, stride, dil, dil))
    out = F.conv2d(out, sd[p + "conv2.weight"], None, 1, dil, dil)
    if (p + "downsample.0.weight") in sd:
        x = F.conv2d(x, sd[p + "downsample.0.weight"], None, stride)
    return F.relu(out + x)


def resnet_feats(x, sd, p="img_extractor.feats.", taps=None):
    """pspnet.py:65-73.  Block 0 of every layer has dilation 1 (pspnet.py:59-62)."""
    x = F.relu(F.conv2d(x, sd[p + "conv1.weight"], None, 2, 3))
    if taps is not None:
        taps["conv1"] = x
    x = F.max_pool2d(x, 3, 2, 1)
    if taps is not None:
        taps["pool"] = x
    for li, (_, blocks, stride, dil) in enumerate(RESNET34_LAYERS, start=1):
        for b in range(blocks):
            x = basic_block(x, sd, f"{p}layer{li}.{b}.", stride if b == 0 else 1, 1 if b == 0 else dil)
        if taps is not None:
            taps[f"layer{li}"] = x
    return x


def psp_module(f, sd, p="img_extractor.psp."):
    """pspnet.py:89-94 — pool -> 1x1 conv (no bias) -> relu -> bilinear(align_corners) -> cat."""
    h, w = f.shape[2:]
    priors = [f]
    for i, s in enumerate(PSP_BINS):
        y = F.adaptive_avg_pool2d(f, (s, s))
        y = F.relu(F.conv2d(y, sd[f"{p}stages.{i}.1.weight"]))
        priors.append(F.interpolate(y, size=(h, w), mode="bilinear", align_corners=True))
    return torch.cat(priors, 1)


def psp_upsample(x, sd, p):
    """pspnet.py:105-107 — x2 bilinear(align_corners=True) -> conv3x3+bias -> PReLU(1)."""
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    x = F.conv2d(x, sd[p + "conv.0.weight"], sd[p + "conv.0.bias"], 1, 1)
    return F.prelu(x, sd[p + "conv.1.weight"])


def pspnet(x, sd, taps=None):
    """pspnet.py:142-158 (Dropout2d = identity in eval)."""
    f = resnet_feats(x, sd, taps=taps)
    p = psp_module(f, sd)
    if taps is not None:
        taps["psp"] = p
    p = psp_upsample(p, sd, "img_extractor.up_1.")
    if taps is not None:
        taps["up_1"] = p
    p = psp_upsample(p, sd, "img_extractor.up_2.")
    if taps is not None:
        taps["up_2"] = p
    p = psp_upsample(p, sd, "img_extractor.up_3.")
    if taps is not None:
        taps["up_3"] = p
    return F.conv2d(p, sd["img_extractor.final.weight"], sd["img_extractor.final.bias"])


# ------------------------------------------------------------------ plane sweep (network_v5.py:378-416)
def homography(src_proj, ref_proj):
    """proj = P_src @ inverse(P_ref) in fp32 (network_v5.py:390-392) -> rot [B,3,3], trans [B,3]."""
    proj = torch.matmul(src_proj, torch.inverse(ref_proj))
    return proj[:, :3, :3].contiguous(), proj[:, :3, 3].contiguous()


def homo_warping(src_fea, src_proj, ref_proj, depth_values):
    B, C, H, W = src_fea.shape
    D = depth_values.shape[1]
    rot, trans = homography(src_proj, ref_proj)
    y, x = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32),
                          indexing="ij")
    xyz = torch.stack((x.reshape(-1), y.reshape(-1), torch.ones(H * W)))[None].repeat(B, 1, 1)
    rot_xyz = torch.matmul(rot, xyz)                                        # [B,3,HW]
    rot_depth_xyz = rot_xyz.unsqueeze(2) * depth_values.view(B, 1, D, 1)   # [B,3,D,HW]
    proj_xyz = rot_depth_xyz + trans.view(B, 3, 1, 1)
    proj_xy = proj_xyz[:, :2] / proj_xyz[:, 2:3]
    gx = proj_xy[:, 0] / ((W - 1) / 2) - 1
    gy = proj_xy[:, 1] / ((H - 1) / 2) - 1
    grid = torch.stack((gx, gy), dim=3)                                     # [B,D,HW,2]
    warped = F.grid_sample(src_fea, grid.view(B, D * H, W, 2), mode="bilinear",
                           padding_mode="zeros", align_corners=False)
    return warped.view(B, C, D, H, W)


# ------------------------------------------------------------------ CostRegNet (network_v5.py:260-291)
def _bn(x, sd, p, per_sample=False):
    """eval mode: running statistics.  per_sample: the as-shipped train-mode BatchNorm3d at the batch size the
    reference always runs (1, interface_v5.py:39-56 never calls .eval(); network_v5.py:17-28): every sample is normalised with
    the biased mean / variance of its own D x H x W volume."""
    shape = (1, -1, 1, 1, 1)
    if per_sample:
        mean = x.mean(dim=(2, 3, 4), keepdim=True)
        var = x.var(dim=(2, 3, 4), unbiased=False, keepdim=True)
        return (x - mean) * torch.rsqrt(var + BN_EPS) * sd[p + "weight"].view(shape) + sd[p + "bias"].view(shape)
    inv = torch.rsqrt(sd[p + "running_var"] + BN_EPS) * sd[p + "weight"]
    return (x - sd[p + "running_mean"].view(shape)) * inv.view(shape) + sd[p + "bias"].view(shape)


def conv3d_bn_relu(x, sd, p, stride, per_sample=False):
    """network_v5.py:22-28 — conv (no bias) -> BN3d -> ReLU."""
    return F.relu(_bn(F.conv3d(x, sd[p + "conv.weight"], None, stride, 1), sd, p + "bn.", per_sample))


def deconv3d_bn_relu(x, sd, p, per_sample=False):
    """network_v5.py:246-252 — ConvTranspose3d(s2,p1,op1) -> BN3d -> ReLU."""
    y = F.conv_transpose3d(x, sd[p + "conv.weight"], None, 2, 1, 1)
    return F.relu(_bn(y, sd, p + "bn.", per_sample))


def cost_reg_net(x, sd, p="cost_regularization.", taps=None, upto_conv11=False, norm_mode=0):
    ps = norm_mode == 1
    c0 = conv3d_bn_relu(x, sd, p + "conv0.", 1, ps)
    c1 = conv3d_bn_relu(c0, sd, p + "conv1.", 2, ps)
    c2 = conv3d_bn_relu(c1, sd, p + "conv2.", 1, ps)
    c3 = conv3d_bn_relu(c2, sd, p + "conv3.", 2, ps)
    c4 = conv3d_bn_relu(c3, sd, p + "conv4.", 1, ps)
    c5 = conv3d_bn_relu(c4, sd, p + "conv5.", 2, ps)
    c6 = conv3d_bn_relu(c5, sd, p + "conv6.", 1, ps)
    u7 = c4 + deconv3d_bn_relu(c6, sd, p + "conv7.", ps)      # skip adds are post-ReLU (:287-289)
    u9 = c2 + deconv3d_bn_relu(u7, sd, p + "conv9.", ps)
    u11 = c0 + deconv3d_bn_relu(u9, sd, p + "conv11.", ps)
    if taps is not None:
        taps.update(c0=c0, c1=c1, c2=c2, c3=c3, c4=c4, c5=c5, c6=c6, u7=u7, u9=u9, u11=u11)
    if upto_conv11:
        return u11
    return F.conv3d(u11, sd[p + "prob.weight"], None, 1, 1)


# ------------------------------------------------------------------ heads
def _mlp1d(x, sd, p, idx, last_act=True):
    for j, i in enumerate(idx):
        x = F.conv1d(x, sd[f"{p}.{i}.weight"], sd[f"{p}.{i}.bias"])
        if last_act or j + 1 < len(idx):
            x = F.relu(x)
    return x


def _mlp(x, sd, p):
    x = F.relu(F.linear(x, sd[p + ".0.weight"], sd[p + ".0.bias"]))
    x = F.relu(F.linear(x, sd[p + ".2.weight"], sd[p + ".2.bias"]))
    return F.linear(x, sd[p + ".4.weight"], sd[p + ".4.bias"])


def ortho6d_to_mat(x_raw, y_raw):
    """rotation_utils.py:18-27 — columns [x y z], y=norm(y_raw), z=norm(x_raw x y), x=y x z."""
    def nrm(v):
        return v / torch.clamp(v.norm(dim=1, keepdim=True), min=1e-8)
    y = nrm(y_raw)
    z = nrm(torch.cross(x_raw, y, dim=1))
    x = torch.cross(y, z, dim=1)
    return torch.stack((x, y, z), dim=2)


def view_heads(feat, fused, choose, depth_values, sd, taps=None, tag="", norm_mode=0):
    """network_v5.py:432-465,485-499 for one view."""
    B, C, H, W = feat.shape
    D = depth_values.shape[1]
    P = choose.shape[1]
    emb = feat.view(B, C, -1)
    nocs_feat = torch.gather(emb, 2, choose.unsqueeze(1).expand(B, C, P))
    nocs_feat = _mlp1d(nocs_feat, sd, "instance_color", (0,))
    nocs = torch.tanh(_mlp1d(nocs_feat, sd, "nocs_head", (0, 2, 4), last_act=False))     # [B,3,P]

    prob_full = cost_reg_net(fused, sd, taps=taps, norm_mode=norm_mode).squeeze(1)                            # [B,D,H,W]
    pre = torch.gather(prob_full.view(B, D, -1), 2, choose.unsqueeze(1).expand(B, D, P))
    prob = F.softmax(pre, dim=1)                                                          # [B,D,P]
    depth = torch.sum(prob * depth_values.view(B, D, 1), 1)                               # [B,P]

    fv = fused.view(B, C * D, -1)
    fg = torch.gather(fv, 2, choose.unsqueeze(1).expand(B, C * D, P)).view(B, C, D, P)
    fg = torch.sum(fg * prob.unsqueeze(1), dim=2)                                         # [B,C,P]

    pts = _mlp1d(nocs, sd, "nocs_pts_mlp", (0, 2))
    pf = torch.cat((fg, pts), dim=1)
    pf = _mlp1d(pf, sd, "pose_mlp1", (0, 2))
    glob = pf.mean(2, keepdim=True)
    pf1 = torch.cat([pf, glob.expand_as(pf)], 1)
    pf2 = _mlp1d(pf1, sd, "pose_mlp2", (0, 2)).mean(2)                                    # [B,256]
    r6 = _mlp(pf2, sd, "rotation_estimator")
    r = ortho6d_to_mat(r6[:, :3].contiguous(), r6[:, 3:].contiguous())
    t = _mlp(pf2, sd, "translation_estimator")
    s = _mlp(pf2, sd, "size_estimator")
    if taps is not None:
        taps.update({tag + "pre": pre, tag + "prob": prob, tag + "fg": fg, tag + "pf2": pf2, tag + "r6": r6})
    return nocs.permute(0, 2, 1).contiguous(), depth, r, t, s


@torch.no_grad()
def adapose_forward(sd, img1, choose1, img2, choose2, P1, P2, depths, taps=None, norm_mode=0):
    """Full forward; same argument order as network_v5.py:418. Returns the 10-entry dict.  norm_mode 1: per-sample BatchNorm3d
    statistics (the as-shipped train-mode behaviour at batch 1, Dropout2d still identity)."""
    t1 = {} if taps is not None else None
    feat1 = pspnet(img1, sd, taps=t1)
    feat2 = pspnet(img2, sd)
    D = depths.shape[1]
    warped2 = homo_warping(feat2, P2, P1, depths)
    fused1 = feat1.unsqueeze(2).repeat(1, 1, D, 1, 1) + warped2
    del warped2
    c1 = {} if taps is not None else None
    n1, d1, r1, tt1, s1 = view_heads(feat1, fused1, choose1, depths, sd, taps=c1, tag="v1_", norm_mode=norm_mode)
    if taps is not None:
        taps.update({"v1_" + k: v for k, v in t1.items()})
        taps.update({("v1_" + k if not k.startswith("v1_") else k): v for k, v in c1.items()})
        taps["feat1"] = feat1
        taps["feat2"] = feat2
        taps["fused1"] = fused1
    del fused1
    warped1 = homo_warping(feat1, P1, P2, depths)
    fused2 = feat2.unsqueeze(2).repeat(1, 1, D, 1, 1) + warped1
    del warped1
    n2, d2, r2, tt2, s2 = view_heads(feat2, fused2, choose2, depths, sd, norm_mode=norm_mode)
    return {"view1_nocs": n1, "view2_nocs": n2, "view1_depth": d1, "view2_depth": d2,
            "view1_r": r1, "view1_t": tt1, "view1_s": s1, "view2_r": r2, "view2_t": tt2, "view2_s": s2}
