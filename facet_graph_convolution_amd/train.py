"""Training / inference drivers: the behaviour of the reference's ``trainNet`` and ``inferNetOld``
(train.py:380-632, 29-144) on the MI355X kernels.  The TensorFlow session plumbing is not reproduced; what is:

  * one iteration = random patch (here: random mesh), 4000 random loss rows (fake rows included), a fresh random
    rotation applied to inputs and ground truth, ONE forward + backward + TF1-Adam update (the reference runs the
    forward three times per iteration, train.py:577,619,620: an artefact, not a contract);
  * the NaN watchdog (train.py:505-506,620-623), the smoothed loss every 50 iterations and the CSV of losses
    (train.py:580-586,629-632), checkpoints every SAVEITER iterations (train.py:551-552) with resume;
  * inference: forward without rotation, un-permute, drop fake rows, normalise twice (train.py:115-121,136).
"""
import os

import numpy as np
import torch

from . import ops, tfckpt
from .net import FacetDenoiser, COST_SAMPLES
from .settings import SAVEITER
from .utils import rand_rotation_matrix


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fn, gt):
        idx = torch.arange(fn.shape[0], dtype=torch.int32, device=fn.device)
        out = ops.angular_loss_fwd(fn, gt, idx)
        ctx.save_for_backward(fn, gt, idx, out)
        return out[0]

    @staticmethod
    def backward(ctx, dloss):
        fn, gt, idx, out = ctx.saved_tensors
        return ops.angular_loss_bwd(fn, gt, idx, out, 1.0) * dloss, None


def faceNormalsLoss(fn, gt_fn):
    """train.py:1272-1294: mean angle in degrees over the rows whose ground truth is not a fake (all-zero) row."""
    return _LossFn.apply(fn.reshape(-1, 3).contiguous(), gt_fn.reshape(-1, 3).contiguous().float())


def save_checkpoint(path, net, iteration):
    """saver.save(sess, path, global_step=iteration) (train.py:551-552,626): weights + Adam moments + step as the
    TensorFlow bundle `<path>-<iteration>.index/.data-00000-of-00001` plus the `checkpoint` state file (tfckpt.py).
    A path ending in ".pt" keeps the same state as one torch file instead."""
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    if not path.endswith(".pt"):
        return tfckpt.save_network(path, net, global_step=iteration)
    P = net.params
    torch.save({"theta": P.theta.cpu(), "m": P.m.cpu(), "v": P.v.cpu(), "step": P.step, "iteration": iteration,
                "multi_scale": net.multi_scale}, path)
    return path


def load_checkpoint(path, net):
    """saver.restore (train.py:79-87,522-534).  path: a TensorFlow checkpoint prefix, its .index file or a directory
    with a `checkpoint` state file - written by the reference or by save_checkpoint - or a ".pt" file.  Returns the
    iteration the checkpoint was taken at."""
    if tfckpt.is_tf_checkpoint(path):
        return tfckpt.load_network(path, net)
    if not os.path.isfile(path):
        raise FileNotFoundError("no checkpoint at %s" % path)
    ck = torch.load(path, map_location="cpu")
    P = net.params
    if ck["theta"].numel() != P.theta.numel():
        raise RuntimeError("checkpoint has %d parameters, network has %d" % (ck["theta"].numel(), P.theta.numel()))
    P.theta.copy_(ck["theta"])
    P.m.copy_(ck["m"])
    P.v.copy_(ck["v"])
    P.step = int(ck["step"])
    return int(ck["iteration"])


def trainNet(trainSet, num_iterations, network_path=None, net_name="net", device="cuda", seed=0, log=print,
             capture=False, validSet=None):
    """train.py:380-632.  trainSet / validSet: dataClasses.TrainingSet.  Returns (net, lossArray [iters/50, 2])."""
    meshes = []
    for i in range(len(trainSet.in_list)):
        net = None
        meshes.append((trainSet.in_list[i], trainSet.adj_list[i], trainSet.gt_list[i]))
    net = FacetDenoiser(device, seed=seed)
    start = 0
    ckpt = os.path.join(network_path, net_name) if network_path else None
    if ckpt:
        # resume from the directory's latest checkpoint if it belongs to this network (train.py:525-533)
        st = tfckpt.get_checkpoint_state(network_path)
        if st and st.model_checkpoint_path:
            split = os.path.basename(st.model_checkpoint_path).split('-')
            if split[0] == net_name:
                load_checkpoint(st.model_checkpoint_path, net)
                start = int(split[1]) if len(split) > 1 and split[1].isdigit() else 0
        elif os.path.exists(ckpt + ".pt"):
            start = load_checkpoint(ckpt + ".pt", net)
    rs = np.random.RandomState(seed + 1)
    evalStepNum = 50
    lossArray = np.zeros([max(num_iterations // evalStepNum, 1), 2])
    bound = -1
    train_loss, train_samp, hasNan = 0.0, 0, False
    valid, last_loss = [], 0.0
    if validSet is not None:
        valid = [(validSet.in_list[i], validSet.adj_list[i], validSet.gt_list[i]) for i in range(len(validSet.in_list))]
    for it in range(num_iterations):
        if ckpt and it % SAVEITER == 0 and it > 0:
            save_checkpoint(ckpt, net, start + it)
        b = rs.randint(len(meshes))
        if b != bound:       # the reference feeds a new patch through feed_dict; here every mesh stays bound in HBM
            x, adjs, gt = meshes[b]
            net.bind_cached(b, x, adjs, gt=gt)
            bound = b
        n0 = meshes[b][0].shape[1]
        samp_it = rs.randint(n0, size=COST_SAMPLES)
        R_it = rand_rotation_matrix(randnums=rs.uniform(size=3))
        if valid and it % (evalStepNum * 2) == 0:
            # train.py:588-617: every 100 iterations the loss alone on every validation mesh, with this iteration's
            # rotation and fresh random rows, BEFORE the training step; the previous row of the CSV gets the mean of
            # this and the last value
            valid_loss = 0.0
            for vbm, (vx, vadj, vgt) in enumerate(valid):
                net.bind_cached(("valid", vbm), vx, vadj, gt=vgt)
                net.set_samples(rs.randint(vx.shape[1], size=COST_SAMPLES))
                net.set_rotation(R_it)
                valid_loss += net.eval_loss(rotate=True)[0].item()
            valid_loss /= len(valid)
            log("Iteration %d, validation loss %g" % (it, valid_loss))
            row = min(it // evalStepNum, len(lossArray) - 1)
            lossArray[row, 1] = valid_loss
            if it > 0:
                lossArray[row - 1, 1] = (valid_loss + last_loss) / 2
                last_loss = valid_loss
            net.bind_cached(b, *meshes[b][:2], gt=meshes[b][2])
        loss = net.train_step(sample_ind=samp_it, R=R_it, capture=capture)
        if it % evalStepNum == 0 or it == num_iterations - 1:
            lv = loss[0].item()      # the only host sync of the loop
            if not np.isfinite(lv):  # NaN watchdog (train.py:620-623)
                hasNan = True
                log("WARNING! NAN FOUND AFTER TRAINING!!!! training example %d/%d" % (b, len(meshes)))
            train_loss += lv
            train_samp += 1
            if it % evalStepNum == 0:
                log("Iteration %d, training loss %g" % (it, train_loss / train_samp))
                lossArray[min(it // evalStepNum, len(lossArray) - 1), 0] = train_loss / train_samp
                train_loss, train_samp = 0.0, 0
    if ckpt:
        save_checkpoint(ckpt, net, start + num_iterations)
        with open(os.path.join(network_path, net_name + ".csv"), "ab") as fh:
            np.savetxt(fh, lossArray, delimiter=",")
    return net, lossArray


def update_position2(x, face_normals, edge_map, v_edges, iter_num=20, max_edges=20):
    """train.py:1467-1557, same argument layout on torch GPU tensors: x [1,V,3], face_normals [1,F,3], edge_map
    int [1,E,4], v_edges int [1,V,max_edges] (-1 = unused slot); returns the updated positions [1,V,3].
    lambda = 1/18 as in the reference (:1469)."""
    from . import ops, tfckpt
    if v_edges.shape[-1] != max_edges:
        raise ValueError("v_edges has %d slots per vertex, max_edges says %d" % (v_edges.shape[-1], max_edges))
    out = ops.vertex_update(x.reshape(-1, 3), face_normals.reshape(-1, 3), edge_map.reshape(-1, 4),
                            v_edges.reshape(-1, max_edges), iter_num)
    return out.unsqueeze(0)


def updateFacesCenter(vertices, faces, coarsening_steps):
    """train.py:1768-1798: node centres of the three levels, [fpos0 [1,N0,3], fpos1 [1,N0/4,3], fpos2 [1,N0/16,3]]."""
    from . import ops, tfckpt
    if coarsening_steps != 2:
        raise NotImplementedError("libfgc pools 4:1 (coarsening_steps = 2, settings.py:31)")
    f0 = ops.face_centers(vertices.reshape(-1, 3), faces.reshape(-1, 3))
    f1 = ops.pool4_avg_iz(f0)
    f2 = ops.pool4_avg_iz(f1)
    return [f0.unsqueeze(0), f1.unsqueeze(0), f2.unsqueeze(0)]


def update_position_MS(x, face_normals_list, faces, v_faces0, coarsening_steps, iter_num_list=[80, 20, 20]):
    """train.py:1668-1764, same arguments on torch GPU tensors: x [1,V,3], face_normals_list = [n0 [1,N0,3],
    n1 [1,N0/4,3], n2 [1,N0/16,3]], faces int [1,N0,3] (fake nodes = -1 rows), v_faces0 int [1,V,K].  Returns
    (x [1,V,3], [dx of the coarse, the middle and the fine stage, each [V,3]])."""
    from . import ops, tfckpt
    if coarsening_steps != 2 or len(face_normals_list) != 3:
        raise NotImplementedError("three levels pooled 4:1, as the network has them (settings.py:31-32)")
    out, dx = ops.vertex_update_ms(x.reshape(-1, 3), [t.reshape(-1, 3) for t in face_normals_list], faces.reshape(-1, 3),
                                   v_faces0.reshape(x.reshape(-1, 3).shape[0], -1), iter_num_list)
    return out.unsqueeze(0), [dx[0], dx[1], dx[2]]


def inferNet(inputMesh, net_or_checkpoint, device="cuda"):
    """train.py:147-376: the multi-scale network (three heads), its three normal fields normalised, update_position_MS
    with [80, 20, 20] iterations - on the whole mesh, or patch by patch when addMeshWithVertices cut it (maxSize).
    inputMesh: InferenceMesh filled by addMeshWithVertices; the network must have been built with multi_scale=True.
    Returns the reference's 9-tuple
    (points, points_mid, points_coarse, fine / mid / coarse normals [F,3] in face order, fine / mid / coarse positions
    [F,3]; as in the reference the three position arrays are the input barycentre channels)."""
    from . import ops, tfckpt
    if isinstance(net_or_checkpoint, FacetDenoiser):
        net = net_or_checkpoint
    else:
        net = FacetDenoiser(device, multi_scale=True)
        load_checkpoint(net_or_checkpoint, net)
    if not net.multi_scale:
        raise ValueError("inferNet needs a network with the multi-scale heads (multi_scale=True)")
    if not inputMesh.v_list or len(inputMesh.v_list) != len(inputMesh.in_list):
        raise ValueError("inferNet needs a mesh prepared by addMeshWithVertices")
    dev = net.device
    c = lambda t: t.cpu().numpy()
    n_patches = len(inputMesh.in_list)
    if n_patches > 1:
        # train.py:255-330: the vertex positions of the patches are averaged over the patches that hold a vertex, the
        # normals of a face are those of the last patch that holds it, and (as in the reference) the three position
        # arrays are the last patch's input barycentre channels, padded and in node order
        vnum, fnum = inputMesh.vNum, inputMesh.fNum
        acc = [torch.zeros(vnum, 3, dtype=torch.float32, device=dev) for _ in range(3)]
        weights = torch.zeros(vnum, 3, dtype=torch.float32, device=dev)
        normals = [torch.zeros(fnum, 3, dtype=torch.float32, device=dev) for _ in range(3)]
        pos = None
    for i in range(n_patches):
        x, adjs = inputMesh.in_list[i], inputMesh.adj_list[i]
        net.bind_mesh(x, adjs)
        net.forward(rotate=False)
        B = net.buffers
        n0 = B["nconv"]                                              # normalizeTensor(y0), done by forward()
        n1 = _normalize_rows_like_reference(B["y1"])                 # train.py:192-193
        n2 = _normalize_rows_like_reference(B["y2"])
        xp = torch.as_tensor(inputMesh.v_list[i][0], dtype=torch.float32, device=dev)
        faces = torch.as_tensor(np.asarray(inputMesh.faces_list[i][0]).astype(np.int32), device=dev)
        vf = torch.as_tensor(np.asarray(inputMesh.v_faces_list[i][0]).astype(np.int32), device=dev)
        pts, dx = ops.vertex_update_ms(xp, [n0, n1, n2], faces, vf, (80, 20, 20))
        pts_mid = pts - dx[2]                                        # train.py:249-250
        pts_coarse = pts_mid - dx[1]
        perm = torch.as_tensor(np.asarray(inputMesh.permutations[i]).astype(np.int64), device=dev)
        nf = inputMesh.num_faces[i]
        up1 = n1.repeat_interleave(4, dim=0)                         # custom_upsampling, train.py:222-223
        up2 = n2.repeat_interleave(16, dim=0)
        fine = n0[perm][:nf]
        mid = _normalize_rows_like_reference(up1)[perm][:nf]
        coarse = _normalize_rows_like_reference(up2)[perm][:nf]
        pos_nodes = torch.as_tensor(np.asarray(x)[0, :, 3:].astype(np.float32), device=dev)
        if n_patches == 1:
            pos = pos_nodes[perm][:nf]                               # train.py:288,296-297
            torch.cuda.synchronize()
            return c(pts), c(pts_mid), c(pts_coarse), c(fine), c(mid), c(coarse), c(pos), c(pos), c(pos)
        vold = torch.as_tensor(np.asarray(inputMesh.vOldInd_list[i]).astype(np.int64), device=dev)
        fold = torch.as_tensor(np.asarray(inputMesh.fOldInd_list[i]).astype(np.int64), device=dev)
        for a, p_ in zip(acc, (pts, pts_mid, pts_coarse)):
            a.index_add_(0, vold, p_)
        weights.index_add_(0, vold, torch.ones_like(pts))
        for nrm, v in zip(normals, (fine, mid, coarse)):
            nrm[fold] = v
        pos = pos_nodes
    w = torch.clamp(weights, min=1.0)
    torch.cuda.synchronize()
    return (c(acc[0] / w), c(acc[1] / w), c(acc[2] / w), c(normals[0]), c(normals[1]), c(normals[2]), c(pos), c(pos),
            c(pos))


def _normalize_rows_like_reference(t):
    """utils.normalizeTensor (utils.py:1700-1715) on [n,3] rows through the library's kernels."""
    from .model import normalizeTensor
    return normalizeTensor(t.unsqueeze(0))[0]


def inferNetOld(inputMesh, net_or_checkpoint, device="cuda", update_vertices=False):
    """train.py:29-144.  inputMesh: dataClasses.InferenceMesh.  Returns the predicted unit normals [F, 3] (numpy, face
    order); with update_vertices=True the reference's full return value (outPoints [V,3], predicted_normals): the
    vertex positions after 60 iterations of update_position2 on those normals (train.py:129-139)."""
    if isinstance(net_or_checkpoint, FacetDenoiser):
        net = net_or_checkpoint
    else:
        net = FacetDenoiser(device)
        load_checkpoint(net_or_checkpoint, net)
    n_patches = len(inputMesh.in_list)
    if n_patches == 1:
        net.bind_mesh(inputMesh.in_list[0], inputMesh.adj_list[0])
        out = net.infer_normals(inputMesh.permutations[0], inputMesh.num_faces[0])
    else:
        # train.py:92-126,136: every patch predicts its own faces (context faces included); predictions of faces
        # covered by several patches are summed in the original face order, then normalised
        F = inputMesh.faces.shape[0]
        acc = torch.zeros(F, 3, dtype=torch.float32, device=net.device)
        for i in range(n_patches):
            net.bind_mesh(inputMesh.in_list[i], inputMesh.adj_list[i])
            n_conv = net.forward(rotate=False)
            perm = torch.as_tensor(np.asarray(inputMesh.permutations[i]).astype(np.int64), device=net.device)
            outN = n_conv[perm][:inputMesh.num_faces[i]]
            idx = torch.as_tensor(np.asarray(inputMesh.patch_indices[i]).astype(np.int64), device=net.device)
            acc.index_add_(0, idx, outN)
        out = acc
        for _ in range(2):          # utils.normalize = normalizeOnce twice
            out = out * (1.0 / (out.norm(dim=1, keepdim=True) + 1e-8))
    if update_vertices:
        if getattr(inputMesh, "edge_map", None) is None:
            raise RuntimeError("the mesh has a vertex with more than MAX_EDGES edges: no edge tables, no vertex update")
        dev = out.device
        xp = torch.as_tensor(inputMesh.vertices, dtype=torch.float32, device=dev)
        pts = update_position2(xp, out.unsqueeze(0), torch.as_tensor(inputMesh.edge_map, device=dev),
                               torch.as_tensor(inputMesh.v_e_map, device=dev), iter_num=60,
                               max_edges=inputMesh.v_e_map.shape[2])
        torch.cuda.synchronize()
        return pts[0].cpu().numpy(), out.cpu().numpy()
    torch.cuda.synchronize()
    return out.cpu().numpy()
