"""The tensor contract between mesh preprocessing and the network, mirroring the reference's
``dataClasses.py`` (PreprocessedData / TrainingSet / InferenceMesh, dataClasses.py:6-27,480-531).

Same attribute names and per-mesh list layout:
    in_list[i]      float64 [1, N0, 6]    unit normal | barycentre / bbox diagonal, fake rows all-zero
    adj_list[i][l]  int64   [1, N_l, 23]  one-indexed K-lists of the 3 graph levels (binary-tree order)
    gt_list[i]      float64 [1, N0, 3]    ground-truth normals (training sets)
    num_faces[i], patch_indices[i], permutations[i] (= inv_perm(newToOld))
    edge_map int32 [1, E, 4], v_e_map int32 [1, V, MAX_EDGES]  (getEdgeMap, for the vertex update)

Differences, by design: the adjacency / coarsening loops run natively (libfgc host routines) and a mesh is
kept whole by default (maxSize = 10^9): the reference cuts meshes above MAX_PATCH_SIZE = 20 000 faces into BFS
patches only because one TF graph could not hold more (settings.py:20-22); a 1M-facet mesh fits one MI355X many times
over.  Constructed with maxSize = MAX_PATCH_SIZE the reference's patch mode is reproduced (same patches for the same
numpy seed; receptive fields truncated at patch borders, overlaps summed at inference).
"""
import numpy as np

from . import utils
from .settings import K_faces, COARSENING_STEPS, COARSENING_LVLS, MIN_PATCH_SIZE, MAX_EDGES


class PreprocessedData(object):
    def __init__(self, maxSize=10 ** 9, coarseningStepNum=COARSENING_STEPS, coarseningLvlNum=COARSENING_LVLS):
        self.in_list = []
        self.gt_list = []
        self.adj_list = []
        self.mesh_count = 0
        self.num_faces = []
        self.patch_indices = []
        self.permutations = []
        self.parents_list = []          # recorded cluster assignments (lets a run be replayed bit-exactly)
        self.maxSize = maxSize
        self.patchSize = maxSize
        self.coarseningStepNum = coarseningStepNum
        self.coarseningLvlNum = coarseningLvlNum
        self.minPatchSize = MIN_PATCH_SIZE
        self.seed = 0
        # multi-scale vertex pipeline (addMeshWithVertices, dataClasses.py:236-456)
        self.v_list = []
        self.faces_list = []
        self.v_faces_list = []
        self.vOldInd_list = []
        self.fOldInd_list = []

    def addMesh_TimeEfficient(self, V0, faces0, GTV=None, seed=None, parents=None):
        """dataClasses.py:34-233: whole mesh (:172-233) or, above maxSize faces, breadth-first patches (:76-171)."""
        V0 = np.asarray(V0, dtype=np.float32)
        faces0 = np.asarray(faces0)
        facesNum = faces0.shape[0]
        # dataClasses.py:40-42: edge tables for the vertex update that follows inference
        try:
            self.edge_map, self.v_e_map = utils.getEdgeMap(faces0, maxEdges=MAX_EDGES)
            self.edge_map = np.expand_dims(self.edge_map, axis=0)
            self.v_e_map = np.expand_dims(self.v_e_map, axis=0)
        except RuntimeError:        # a vertex with more than MAX_EDGES edges: the reference dies here with an
            self.edge_map = self.v_e_map = None   # IndexError; normals can still be inferred, only the vertex update cannot
        f_normals0, f_pos0 = utils.face_features(V0, faces0)
        f_adj0 = utils.getFacesLargeAdj(faces0, K_faces)
        f_normals_pos = np.concatenate((f_normals0, f_pos0), axis=1)        # float64 [F,6] (dataClasses.py:64)
        GTf_normals0 = utils.computeFacesNormals(GTV, faces0) if GTV is not None else None
        if facesNum > self.maxSize:
            # dataClasses.py:76-171: breadth-first patches of patchSize faces (already covered faces join as context),
            # seeds drawn with numpy's global RNG among the uncovered faces unless the previous patch proposed one
            print("Dividing mesh into patches: %i faces (%i max allowed)" % (facesNum, self.maxSize))
            faceCheck = np.zeros(facesNum)
            faceRange = np.arange(facesNum)
            nextSeed = -1
            while np.any(faceCheck == 0):
                toBeProcessed = faceRange[faceCheck == 0]
                faceSeed = toBeProcessed[np.random.randint(toBeProcessed.shape[0])] if nextSeed == -1 else nextSeed
                patchAdj, fOldInd, nextSeed = utils.getGraphPatch_wMask(f_adj0, self.patchSize, faceSeed, faceCheck,
                                                                        self.minPatchSize)
                faceCheck[fOldInd] = 1
                if fOldInd.shape[0] < 100:      # small disjoint components are dropped (dataClasses.py:103-104)
                    continue
                self._add_graph(patchAdj, f_normals_pos[fOldInd],
                                GTf_normals0[fOldInd] if GTf_normals0 is not None else None, fOldInd, seed, parents)
        else:
            self._add_graph(f_adj0, f_normals_pos, GTf_normals0, [], seed, parents)

    def addMeshWithVertices(self, V0, faces0, GTV=None, seed=None, parents=None):
        """dataClasses.py:236-456, whole-mesh branch (:374-440): what the multi-scale network + update_position_MS
        consume.  On top of addMesh_TimeEfficient: the faces padded with -1 rows for the fake nodes and re-ordered like
        the nodes (faces_list), the node ids of the faces around every vertex (v_faces_list, 25 slots) and the vertices
        divided by the bounding-box diagonal (v_list).  Returns (vNum, facesNum)."""
        V0 = np.asarray(V0, dtype=np.float32)
        faces0 = np.asarray(faces0)
        if faces0.shape[0] > self.maxSize:
            return self._add_mesh_patches_with_vertices(V0, faces0, GTV, seed)
        first = len(self.in_list)
        self.addMesh_TimeEfficient(V0, faces0, GTV=GTV, seed=seed, parents=parents)
        oldToNew = self.permutations[first]
        new_N = len(oldToNew)
        newToOld = np.empty(new_N, dtype=np.int64)
        newToOld[oldToNew] = np.arange(new_N)
        faces_p = np.concatenate((faces0.astype(np.int64), -np.ones((new_N - faces0.shape[0], 3), dtype=np.int64)),
                                 axis=0)[newToOld]
        v_faces = utils.getVerticesFaces(faces_p, 25, V0.shape[0])
        if GTV is not None:
            Vn, GTn = utils.normalizePointSets(V0, np.asarray(GTV, dtype=np.float32))
            self.__dict__.setdefault("gtv_list", []).append(GTn[np.newaxis])
        else:
            Vn, _ = utils.normalizePointSets(V0, V0)
        self.v_list.append(Vn[np.newaxis])
        self.faces_list.append(faces_p[np.newaxis])
        self.v_faces_list.append(v_faces[np.newaxis])
        self.fOldInd_list.append([])
        self.vOldInd_list.append([])
        return V0.shape[0], faces0.shape[0]

    def _add_mesh_patches_with_vertices(self, V0, faces0, GTV, seed):
        """dataClasses.py:270-372: meshes above maxSize are cut into breadth-first MESH patches (getMeshPatch) around
        seeds drawn with np.random among the faces no patch has covered yet, until every face is covered; components of
        fewer than 100 faces are dropped, and with a ground-truth point cloud so are patches whose bounding box holds
        fewer GT points than the patch has vertices.  Every patch brings its own vertices (normalised coordinates),
        faces, vertex-face table and graph levels; vOldInd_list / fOldInd_list map them back."""
        f_normals0 = utils.computeFacesNormals(V0, faces0)
        f_adj0 = utils.getFacesLargeAdj(faces0, K_faces)
        f_pos0 = utils.getTrianglesBarycenter(V0, faces0, normalize=True)
        f_normals_pos = np.concatenate((f_normals0, f_pos0), axis=1)
        addGT = GTV is not None
        if addGT:
            GTV = np.asarray(GTV, dtype=np.float32)
            gtf_normals0 = utils.computeFacesNormals(GTV, faces0)
            Vn, GTn = utils.normalizePointSets(V0, GTV)
        else:
            Vn, _ = utils.normalizePointSets(V0, V0)
        facesNum = faces0.shape[0]
        faceCheck = np.zeros(facesNum)
        faceRange = np.arange(facesNum)
        while np.any(faceCheck == 0):
            toBeProcessed = faceRange[faceCheck == 0]
            faceSeed = toBeProcessed[np.random.randint(toBeProcessed.shape[0])]
            pV, pF, pAdj, vOldInd, fOldInd = utils.getMeshPatch(Vn, faces0, f_adj0, self.patchSize, faceSeed)
            faceCheck[fOldInd] += 1
            if fOldInd.shape[0] < 100:          # small disjoint components are not added
                continue
            if addGT:
                patchGTV = utils.takePointSetSlice(GTn, utils.getBoundingBox(pV))
                if patchGTV.shape[0] < pV.shape[0]:     # no ground truth in the window: a fake surface
                    continue
            first = len(self.in_list)
            self._add_graph(pAdj, f_normals_pos[fOldInd], gtf_normals0[fOldInd] if addGT else None, fOldInd, seed, None)
            oldToNew = self.permutations[first]
            new_N = len(oldToNew)
            newToOld = np.empty(new_N, dtype=np.int64)
            newToOld[oldToNew] = np.arange(new_N)
            faces_p = np.concatenate((pF.astype(np.int64), -np.ones((new_N - pF.shape[0], 3), dtype=np.int64)),
                                     axis=0)[newToOld]
            self.v_list.append(pV[np.newaxis])
            self.faces_list.append(faces_p[np.newaxis])
            self.v_faces_list.append(utils.getVerticesFaces(faces_p, 25, pV.shape[0])[np.newaxis])
            self.vOldInd_list.append(vOldInd)
            self.fOldInd_list.append(fOldInd)
            if addGT:
                self.__dict__.setdefault("gtv_list", []).append(patchGTV[np.newaxis])
        return V0.shape[0], facesNum

    def _add_graph(self, f_adj, f_normals_pos, GTf_normals, patch_index, seed, parents):
        """One mesh or one patch: coarsen, pad with fake nodes, reorder, append (dataClasses.py:106-171,172-233)."""
        old_N = f_adj.shape[0]
        f_normals0, f_pos0 = f_normals_pos[:, :3], f_normals_pos[:, 3:]
        GTf_normals0 = GTf_normals
        if self.coarseningLvlNum > 1:
            levels = (self.coarseningLvlNum - 1) * self.coarseningStepNum
            keep = tuple(self.coarseningStepNum * l for l in range(self.coarseningLvlNum))
            cur_seed = self.seed if seed is None else seed
            has_sat = True
            while has_sat:      # dataClasses.py:179-192: re-draw the pairing while a row saturates K
                klists, newToOld, par, has_sat = utils.coarsen_klists(f_adj, f_pos0, f_normals0, levels, K_faces,
                                                                      cur_seed, parents, keep)
                if has_sat and parents is not None:
                    raise RuntimeError("recorded cluster assignments saturate K=%d" % K_faces)
                cur_seed += 1
            self.seed = cur_seed
            fAdjs = [k[np.newaxis].astype(np.int64) for k in klists]
            new_N = len(newToOld)
            pad6 = np.zeros((new_N - old_N, f_normals_pos.shape[1]))
            f_normals_pos = np.concatenate((f_normals_pos, pad6), axis=0)[newToOld]
            if GTf_normals0 is not None:
                GTf_normals0 = np.concatenate((GTf_normals0, np.zeros((new_N - old_N, 3))), axis=0)[newToOld]
            self.parents_list.append(par)
        else:
            fAdjs = [f_adj[np.newaxis].astype(np.int64)]
            newToOld = None
        self.num_faces.append(old_N)
        self.patch_indices.append(patch_index)
        if newToOld is not None:
            self.permutations.append(utils.inv_perm(newToOld))
        self.in_list.append(f_normals_pos[np.newaxis])
        self.adj_list.append(fAdjs)
        if GTf_normals0 is not None:
            self.gt_list.append(np.asarray(GTf_normals0, dtype=np.float64)[np.newaxis])
        self.mesh_count += 1


class TrainingSet(PreprocessedData):
    """dataClasses.py:480-506 (array interface; OBJ parsing is outside the hot path)."""

    def addMeshWithGT(self, V_noisy, faces, V_gt, seed=None, parents=None):
        """dataClasses.py:490-494.  Either the reference's call addMeshWithGT(inputFilePath, filename, gtFilePath,
        gtfilename) (both OBJs are read with utils.load_mesh; the faces of the noisy file are used for both) or arrays
        addMeshWithGT(V_noisy, faces, V_gt)."""
        if isinstance(V_noisy, str):
            in_path, in_name, gt_path, gt_name = V_noisy, faces, V_gt, seed
            V_noisy, _, _, faces, _ = utils.load_mesh(in_path, in_name, 0, False)
            V_gt = utils.load_mesh(gt_path, gt_name, 0, False)[0]
            seed = None
        self.addMesh_TimeEfficient(V_noisy, faces, GTV=V_gt, seed=seed, parents=parents)


class InferenceMesh(PreprocessedData):
    """dataClasses.py:510-531."""

    def addMesh(self, V, faces, seed=None, parents=None):
        """dataClasses.py:513-519.  Either the reference's call addMesh(inputFilePath, filename) (an OBJ is read with
        utils.load_mesh) or arrays addMesh(V, faces)."""
        if isinstance(V, str):
            V, _, _, faces, _ = utils.load_mesh(V, faces, 0, False)
        self.vertices = np.asarray(V, dtype=np.float32)[np.newaxis]
        self.faces = np.asarray(faces)
        self.addMesh_TimeEfficient(V, faces, seed=seed, parents=parents)
        self.normals = utils.computeFacesNormals(self.vertices[0], self.faces)

    def addMeshWithVertices(self, V, faces=None, seed=None, parents=None):
        """dataClasses.py:521-531 (file form addMeshWithVertices(inputFilePath, filename), or arrays)."""
        if isinstance(V, str):
            V, _, _, faces, _ = utils.load_mesh(V, faces, 0, False)
        V = np.asarray(V, dtype=np.float32)
        self.fNum, self.vNum = np.asarray(faces).shape[0], V.shape[0]
        PreprocessedData.addMeshWithVertices(self, V, faces, seed=seed, parents=parents)
        self.vertices = V[np.newaxis]
        self.faces = np.asarray(faces)
        self.normals = utils.computeFacesNormals(V, self.faces)
        return self.vNum, self.fNum
