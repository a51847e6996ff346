"""ORACLE (test infrastructure, never shipped, never measured as the product).

CPU restatement, in torch, of the reference's graph-conv denoising network.  It keeps the
reference's *op sequence* (zero-row pad -> K-padded gather -> per-edge softmax -> multiply
-> reduce over K and M -> masked bias), so it is "reference-shaped": its cost profile and
its fp32 summation structure are those of ``/root/reference/Code/model.py``, and torch
autograd through it is the gradient oracle.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product (``facet_graph_convolution_amd``) must not.

Pinned by: tests/test_oracle_golden.py against tests/golden/{conv_*,net_*,infer_*}.npz,
which were produced by executing the reference source (tests/golden/gen/make_golden.py).
Parity against real TensorFlow *binaries* is unpinned (TF is not installable offline).

Reference lines followed are cited per function as ``ref: file:line``.
"""
import math

import numpy as np
import torch

K_FACES = 23          # ref: settings.py:23
M_ASSIGN = 9          # ref: model.py:855,868,880
LRELU_ALPHA = 0.1     # ref: model.py:846 (overrides :843)
STD_W = 0.05          # ref: model.py:17
STD_B = 0.01          # ref: model.py:18


# --------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------
def param_spec(multi_scale=False, in_channels=6):
    """(kind, shape) of every trainable variable in creation order.

    ref: custom_conv2d creates W0[M,Cout,Cin], b[Cout], u[M,Cin], c[M], v[M,Cin]
    (model.py:430-433,447); custom_lin creates W[in,out], b[out] (model.py:767-768);
    layer order from get_model_reg_multi_scale (model.py:855-941).  With multiScale the
    two head MLPs are created after dconv3 and after dconv2 (model.py:894-899,915-920).
    """
    M = M_ASSIGN

    def conv(cin, cout):
        return [("weight", (M, cout, cin)), ("bias", (cout,)), ("assignment", (M, cin)),
                ("assignment", (M,)), ("assignment", (M, cin))]

    def lin(cin, cout):
        return [("weight", (cin, cout)), ("bias", (cout,))]

    spec = []
    spec += conv(in_channels, 32)     # conv1   L0
    spec += conv(32, 64)              # conv2   L1
    spec += conv(64, 128)             # conv3   L2
    spec += conv(128, 128)            # dconv3  L2
    if multi_scale:
        spec += lin(128, 1024) + lin(1024, 3)
    spec += conv(128, 64)             # upconv2 L1
    spec += conv(128, 64)             # dconv2  L1
    if multi_scale:
        spec += lin(64, 1024) + lin(1024, 3)
    spec += conv(64, 32)              # upconv1 L0
    spec += conv(64, 32)              # dconv1  L0
    spec += lin(32, 1024) + lin(1024, 3)
    return spec


def init_params(seed=0, multi_scale=False, in_channels=6, dtype=torch.float32):
    """Seeded N(0, std) values in creation order (std: model.py:16-44).  Same stream as the
    fixture generator's ``param_values``."""
    rs = np.random.RandomState(seed)
    out = []
    for kind, shape in param_spec(multi_scale, in_channels):
        std = STD_B if kind == "bias" else STD_W
        out.append(torch.tensor(rs.normal(0.0, std, size=shape).astype(np.float32), dtype=dtype))
    return out


def conv_params(cin, cout, seed, M=M_ASSIGN, dtype=torch.float32):
    rs = np.random.RandomState(seed)
    shapes = [("weight", (M, cout, cin)), ("bias", (cout,)), ("assignment", (M, cin)),
              ("assignment", (M,)), ("assignment", (M, cin))]
    return [torch.tensor(rs.normal(0.0, STD_B if k == "bias" else STD_W, size=s).astype(np.float32), dtype=dtype)
            for k, s in shapes]


# --------------------------------------------------------------------------------------
# ops
# --------------------------------------------------------------------------------------
def get_patches(x, adj):
    """ref: model.py:380-405.  x [B,n,C], adj [B,n,K] one-indexed; index 0 -> zero row."""
    B, n, C = x.shape
    xp = torch.cat([torch.zeros(B, 1, C, dtype=x.dtype), x], dim=1)
    return torch.stack([xp[b][adj[b].long()] for b in range(B)], dim=0)  # [B,n,K,C]


def get_weight_assignments(x, adj, u, v, c):
    """ref: model.py:74-95.  q[b,i,k,:] = softmax_m(u x_i + v x_j + c)."""
    ux = torch.matmul(x, u.t())                      # [B,n,M]
    vx = torch.matmul(x, v.t())                      # [B,n,M]
    patches = get_patches(vx, adj)                   # [B,n,K,M]
    logits = ux.unsqueeze(2) + patches + c
    return torch.softmax(logits, dim=-1)


def custom_conv2d(x, adj, params, biasMask=True):
    """ref: model.py:427-504 (invariance-off branch).  params = (W0,b,u,c,v)."""
    W0, b, u, c, v = params
    M, cout, cin = W0.shape
    B, n, _ = x.shape
    K = adj.shape[2]
    adj_size = (adj != 0).sum(dim=2)                                  # :436
    non_zeros = adj_size != 0                                         # :438
    adj_size_f = adj_size.to(x.dtype)
    inv = torch.where(non_zeros, 1.0 / adj_size_f, torch.zeros_like(adj_size_f))  # :440
    W = W0.reshape(M * cout, cin)                                     # :464
    wx = torch.matmul(x, W.t())                                       # :466-468  [B,n,M*cout]
    patches = get_patches(wx, adj)                                    # :470      [B,n,K,M*cout]
    q = get_weight_assignments(x, adj, u, v, c)                       # :474      [B,n,K,M]
    patches = patches.reshape(B, n, K, M, cout)                       # :482
    patches = q.unsqueeze(-1) * patches                               # :484-486
    patches = patches.sum(dim=2)                                      # :488      [B,n,M,cout]
    patches = inv.reshape(B, n, 1, 1) * patches                       # :491
    patches = patches.sum(dim=2)                                      # :493
    if biasMask:
        patches = torch.where(non_zeros.unsqueeze(-1), patches + b, patches)  # :498
    else:
        patches = patches + b
    return patches


def custom_lin(x, W, b):
    """ref: model.py:763-769."""
    return torch.matmul(x, W) + b


def lrelu(x, alpha=LRELU_ALPHA):
    """ref: model.py:828-830."""
    return torch.relu(x) - alpha * torch.relu(-x)


def custom_binary_tree_pooling(x, steps=2):
    """ref: model.py:779-788 ('max').  amax spreads the gradient evenly over ties like tf.reduce_max."""
    B, n, C = x.shape
    return x.reshape(B, -1, 2 ** steps, C).amax(dim=2)


def custom_upsampling(x, steps=2):
    """ref: model.py:817-825."""
    B, n, C = x.shape
    return x.unsqueeze(2).repeat(1, 1, 2 ** steps, 1).reshape(B, -1, C)


def get_model_reg_multi_scale(x, adjs, params, multiScale=False):
    """ref: model.py:837-946.  params: flat list in creation order (see param_spec)."""
    p = list(params)
    pos = [0]

    def take(k):
        r = p[pos[0]:pos[0] + k]
        pos[0] += k
        return r

    a = LRELU_ALPHA
    h_conv1_act = lrelu(custom_conv2d(x, adjs[0], take(5)), a)                   # :858-859
    pool1 = custom_binary_tree_pooling(h_conv1_act, 2)                           # :863
    h_conv2_act = lrelu(custom_conv2d(pool1, adjs[1], take(5)), a)               # :870-871
    pool2 = custom_binary_tree_pooling(h_conv2_act, 2)                           # :875
    h_conv3_act = lrelu(custom_conv2d(pool2, adjs[2], take(5)), a)               # :882-883
    dconv3_act = lrelu(custom_conv2d(h_conv3_act, adjs[2], take(5)), a)          # :890-891
    if multiScale:
        W1, b1, W2, b2 = take(4)
        y_conv2 = custom_lin(lrelu(custom_lin(dconv3_act, W1, b1), a), W2, b2)   # :894-899
    upsamp2 = custom_upsampling(dconv3_act, 2)                                   # :902
    upconv2 = custom_conv2d(upsamp2, adjs[1], take(5))                           # :905 (no activation)
    concat2 = torch.cat([upconv2, h_conv2_act], dim=-1)                          # :909
    dconv2_act = lrelu(custom_conv2d(concat2, adjs[1], take(5)), a)              # :911-912
    if multiScale:
        W1, b1, W2, b2 = take(4)
        y_conv1 = custom_lin(lrelu(custom_lin(dconv2_act, W1, b1), a), W2, b2)   # :915-920
    upsamp1 = custom_upsampling(dconv2_act, 2)                                   # :923
    upconv1 = custom_conv2d(upsamp1, adjs[0], take(5))                           # :926 (no activation)
    concat1 = torch.cat([upconv1, h_conv1_act], dim=-1)                          # :929
    dconv1_act = lrelu(custom_conv2d(concat1, adjs[0], take(5)), a)              # :931-932
    W1, b1, W2, b2 = take(4)
    y_conv0 = custom_lin(lrelu(custom_lin(dconv1_act, W1, b1), a), W2, b2)       # :937-941
    assert pos[0] == len(p)
    if multiScale:
        return y_conv0, y_conv1, y_conv2
    return y_conv0


def normalizeTensor(x):
    """ref: utils.py:1700-1715."""
    eps = torch.tensor(1e-5, dtype=x.dtype)
    x = x / (x.abs().mean() + eps)
    norm = torch.sqrt(eps + (x * x).sum(dim=-1))
    inv = torch.where(norm > eps, 1.0 / (norm + eps), torch.zeros_like(norm))
    return x * inv.unsqueeze(-1)


def faceNormalsLoss(fn, gt_fn):
    """ref: train.py:1272-1294."""
    n_dt = (fn * gt_fn).sum(dim=-1)
    close_to_one = torch.tensor(0.9999999, dtype=fn.dtype)
    loss = torch.acos(torch.minimum(torch.maximum(n_dt, -close_to_one), close_to_one))
    fake = gt_fn.abs().sum(dim=2) <= 10e-4
    real = torch.where(fake, torch.zeros_like(loss), torch.ones_like(loss))
    loss = 180 * loss / math.pi
    loss = torch.where(fake, torch.zeros_like(loss), loss)
    return loss.sum() / real.sum()


def rotate_inputs(x, gt, R):
    """ref: train.py:439-451.  R [3,3] applied to GT normals and to both 3-vectors of each input row."""
    R = R.to(x.dtype)
    gt_r = torch.matmul(gt, R.t()) if gt is not None else None
    B, n, C = x.shape
    x_r = torch.matmul(x.reshape(B, n, C // 3, 3), R.t()).reshape(B, n, C)
    return x_r, gt_r


def train_loss(x, adjs, gt, params, sample_ind, R):
    """ref: train.py:439-517: rotate, net, normalise, sample 4000 rows, angular loss."""
    x_r, gt_r = rotate_inputs(x, gt, R)
    y = get_model_reg_multi_scale(x_r, adjs, params)
    n_conv = normalizeTensor(y)
    idx = torch.as_tensor(sample_ind, dtype=torch.long)
    return faceNormalsLoss(n_conv[:, idx], gt_r[:, idx]), n_conv


def pooled_gt(gt):
    """NOT in the reference (build extension, facet_graph_convolution_amd.net multi-scale training): ground truth of the
    coarse heads = the fine normals pooled with avg_ignore_zeros (ref: model.py:792-814) and renormalised; rows that pool
    only fake nodes stay zero."""
    g = avg_ignore_zeros_pool(gt, 2)
    nrm = g.norm(dim=-1, keepdim=True)
    return torch.where(nrm > 0, g / nrm.clamp_min(1e-20), torch.zeros_like(g))


def train_loss_ms(x, adjs, gt, params, sample_ind, R):
    """NOT in the reference (its multi-scale training runs a point-set loss through the vertex update, train.py:1075-1105):
    the three heads of get_model_reg_multi_scale(multiScale=True), each through normalizeTensor and faceNormalsLoss against
    the pooled ground truth on the sampled rows modulo the level's size.  Returns (loss0 + loss1 + loss2, [loss_k])."""
    x_r, gt_r = rotate_inputs(x, gt, R)
    ys = get_model_reg_multi_scale(x_r, adjs, params, multiScale=True)
    gts = [gt_r, None, None]
    g = gt
    for k in (1, 2):
        g = pooled_gt(g)
        gts[k] = torch.matmul(g, R.to(x.dtype).t())
    idx = torch.as_tensor(sample_ind, dtype=torch.long)
    losses = []
    for y, g in zip(ys, gts):
        n_conv = normalizeTensor(y)
        ik = idx % y.shape[1]
        losses.append(faceNormalsLoss(n_conv[:, ik], g[:, ik]))
    return losses[0] + losses[1] + losses[2], losses


def adam_step_tf1(params, grads, m, v, t, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """TensorFlow-1 Adam (train.py:520 uses tf.train.AdamOptimizer() defaults).

    The algorithm lives in TensorFlow (absent dependency, version unpinned by the
    reference); restated from its documented update rule:
        lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2
        theta -= lr_t * m / (sqrt(v) + eps)          (eps is NOT bias-corrected)
    parity unpinned (no TF binary, no reference fixture).  In-place on the lists; t is 1-based.
    """
    lr_t = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    for p, g, mi, vi in zip(params, grads, m, v):
        mi.mul_(b1).add_(g, alpha=1.0 - b1)
        vi.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        p.sub_(lr_t * mi / (vi.sqrt() + eps))


def infer_epilogue(n_conv, permutations, num_faces):
    """ref: train.py:115-121,136 + utils.py:26-35.  [1,N0,3] -> [F,3] in original face order."""
    outN = n_conv.squeeze(0)[torch.as_tensor(np.asarray(permutations), dtype=torch.long)][:num_faces]
    for _ in range(2):
        norms = torch.sqrt((outN * outN).sum(1, keepdim=True)) + 0.00000001
        outN = outN * (1 / norms)
    return outN


def update_position2(x, face_normals, edge_map, v_edges, iter_num=20, lmbd=1.0 / 18):
    """ref: train.py:1467-1557.  x [V,3], face_normals [F,3], edge_map [E,4] = [v1,v2,f1,f2 (-1: none)],
    v_edges [V,max_edges] (-1: unused).  Per iteration, simultaneously for every vertex i,
        x_i += lmbd * sum over its edges e = (v1, v2, f1, f2) and both faces f of  n_f (n_f . (x_v1 - x_i))
                                                                                 + n_f (n_f . (x_v2 - x_i))
    (the endpoint that is i itself contributes zero; missing faces and unused slots have zero normals)."""
    x = torch.as_tensor(x)
    dt = x.dtype
    fn = torch.cat([torch.zeros(1, 3, dtype=dt), torch.as_tensor(face_normals).to(dt)], 0)      # :1490
    em = torch.as_tensor(np.asarray(edge_map)).long() + torch.tensor([0, 0, 1, 1])                # :1482-1483
    em = torch.cat([torch.zeros(1, 4, dtype=torch.long), em], 0)                                  # :1486-1487
    ne = em[torch.as_tensor(np.asarray(v_edges)).long() + 1]                                      # [V, E, 4]  :1479,1494
    nrm = fn[ne[:, :, 2:]]                                                                        # [V, E, 2, 3]
    nrm4 = torch.cat([nrm, nrm], 2)                                                               # rows n1 n2 n1 n2  :1508
    for _ in range(iter_num):
        pairs = x[ne[:, :, :2]]                                                                   # [V, E, 2, 3]
        d = pairs - x[:, None, None, :]                                                           # :1530
        d4 = torch.stack([d[:, :, 0], d[:, :, 0], d[:, :, 1], d[:, :, 1]], 2)                     # rows d0 d0 d1 d1  :1533-1536
        dp = (d4 * nrm4).sum(-1, keepdim=True)                                                    # :1539
        upd = (nrm4 * dp).sum(2).sum(1)                                                           # :1545-1550
        x = x + lmbd * upd
    return x


def avg_ignore_zeros_pool(x, steps=2):
    """ref: model.py:792-814 (custom_binary_tree_pooling, pooltype='avg_ignore_zeros').  x [B, n, C]; per step pairs of
    consecutive rows are averaged, a row that is zero in EVERY channel being replaced by its partner first (so a fake
    node does not drag the average; two zero rows give zero)."""
    px = x
    for _ in range(steps):
        B, n, C = px.shape
        px = px.reshape(B, n // 2, 2, C)
        l0, l1 = px[:, :, 0], px[:, :, 1]
        z0 = (l0 == 0).all(-1, keepdim=True)
        z1 = (l1 == 0).all(-1, keepdim=True)
        c0 = torch.where(z0, l1, l0)
        c1 = torch.where(z1, l0, l1)
        px = torch.stack([c0, c1], 2).mean(2)
    return px


def update_faces_center(vertices, faces, coarsening_steps=2):
    """ref: train.py:1768-1798.  vertices [V,3], faces [N0,3] int (-1 = fake face: its corners read a zero vertex).
    Returns [fpos0 [1,N0,3], fpos1 [1,N0/4,3], fpos2 [1,N0/16,3]]."""
    vz = torch.cat([torch.zeros(1, 3, dtype=vertices.dtype), vertices], 0)
    fpos0 = vz[torch.as_tensor(np.asarray(faces)).long() + 1].mean(1).unsqueeze(0)
    fpos1 = avg_ignore_zeros_pool(fpos0, coarsening_steps)
    fpos2 = avg_ignore_zeros_pool(fpos1, coarsening_steps)
    return [fpos0, fpos1, fpos2]


def update_position_MS(x, face_normals_list, faces, v_faces0, coarsening_steps=2, iter_num_list=(80, 20, 20)):
    """ref: train.py:1668-1764.  x [V,3]; face_normals_list = [n0 [N0,3], n1 [N0/4,3], n2 [N0/16,3]]; faces [N0,3];
    v_faces0 [V,K] face (node) ids of every vertex at the finest level, -1 padded.  Coarse to fine: at scale s a
    vertex is pulled towards the planes of the level-s nodes above its faces,
        x_v += (1/#faces(v)) * sum_k n (n . (c - x_v)),   n, c = normal / centre of node floor(v_faces0[v,k] / 4^s),
    the centres being recomputed from the current vertices in every iteration.  Returns (x [V,3], [dx per scale])."""
    x = torch.as_tensor(x)
    dt = x.dtype
    vf0 = torch.as_tensor(np.asarray(v_faces0)).long()
    numf = (vf0 != -1).sum(-1).to(dt)
    lmbd = (1.0 / numf).reshape(-1, 1)
    nscale = len(face_normals_list)
    dx_list = []
    for s in range(nscale):
        cur = nscale - 1 - s
        fn = torch.cat([torch.zeros(1, 3, dtype=dt), torch.as_tensor(face_normals_list[cur]).to(dt).reshape(-1, 3)], 0)
        div = int((2 ** coarsening_steps) ** cur)
        vf = torch.div(vf0, div, rounding_mode="floor") + 1          # -1 stays -1 (Python-2 division), then 0
        v_fn = fn[vf]                                                # [V, K, 3]
        x_init = x
        for _ in range(iter_num_list[s]):
            fpos = update_faces_center(x, faces, coarsening_steps)[cur].reshape(-1, 3)
            fpos = torch.cat([torch.zeros(1, 3, dtype=dt), fpos], 0)
            e = fpos[vf] - x[:, None, :]
            n_w = (v_fn * e).sum(-1, keepdim=True)
            x = x + lmbd * (n_w * v_fn).sum(1)
        dx_list.append(x - x_init)
    return x, dx_list
