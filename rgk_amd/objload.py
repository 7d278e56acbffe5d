"""From-scratch Wavefront OBJ/MTL reader (assimp is not available).

Produces, per (object, material) group, what the reference reads from an aiMesh in
Scene::LoadAiMesh (src/scene.cpp:130-208): positions, normals, one uv set, tangents
and triangle indices, after the post-process steps requested in
src/config.cpp:196-228: Triangulate (fan), GenNormals (flat, when the file has none)
or GenSmoothNormals, JoinIdenticalVertices, FindDegenerates (drop zero-area faces'
points/lines), CalcTangentSpace (per-face tangent, Gram-Schmidt against the vertex
normal, averaged over joined vertices).  UNPINNED against assimp.
"""
import os

import numpy as np

f32 = np.float32


def parse_mtl(path):
    mtls, cur = {}, None
    if not os.path.exists(path):
        return mtls
    for line in open(path, errors="replace"):
        p = line.split()
        if not p or p[0].startswith("#"):
            continue
        k = p[0]
        if k == "newmtl":
            cur = {}
            mtls[" ".join(p[1:])] = cur
        elif cur is None:
            continue
        elif k in ("Kd", "Ks", "Ke", "Ka"):
            cur[k] = tuple(float(x) for x in p[1:4])
        elif k in ("Ns", "Ni", "d"):
            cur[k] = float(p[1])
        elif k in ("map_Kd", "map_Ks", "map_Ka"):
            cur[k] = p[-1]
        elif k.lower() in ("map_bump", "bump"):
            cur["map_Bump"] = p[-1]
    return mtls


def load_obj_file(path, smooth_normals=False):
    V, VT, VN = [], [], []
    groups = {}  # material -> list of faces; face = list of (v, vt, vn)
    order = []
    cur = "DefaultMaterial"
    mtls = {}
    for line in open(path, errors="replace"):
        p = line.split()
        if not p:
            continue
        k = p[0]
        if k == "v":
            V.append([float(x) for x in p[1:4]])
        elif k == "vt":
            VT.append([float(p[1]), float(p[2]) if len(p) > 2 else 0.0])
        elif k == "vn":
            VN.append([float(x) for x in p[1:4]])
        elif k == "usemtl":
            cur = " ".join(p[1:])
        elif k == "mtllib":
            mtls.update(parse_mtl(os.path.join(os.path.dirname(path), " ".join(p[1:]))))
        elif k == "f":
            idx = []
            for tok in p[1:]:
                q = tok.split("/")
                vi = int(q[0])
                ti = int(q[1]) if len(q) > 1 and q[1] else 0
                ni = int(q[2]) if len(q) > 2 and q[2] else 0
                vi = vi - 1 if vi > 0 else len(V) + vi
                ti = (ti - 1 if ti > 0 else len(VT) + ti) if ti != 0 else -1
                ni = (ni - 1 if ni > 0 else len(VN) + ni) if ni != 0 else -1
                idx.append((vi, ti, ni))
            if len(idx) < 3:
                continue
            if cur not in groups:
                groups[cur] = []
                order.append(cur)
            for j in range(1, len(idx) - 1):  # fan triangulation
                groups[cur].append((idx[0], idx[j], idx[j + 1]))
    V = np.array(V, dtype=f32).reshape(-1, 3)
    VT = np.array(VT, dtype=f32).reshape(-1, 2)
    VN = np.array(VN, dtype=f32).reshape(-1, 3)
    meshes = []
    for mat in order:
        faces = groups[mat]
        nf = len(faces)
        corner = np.array(faces, dtype=np.int64).reshape(nf * 3, 3)
        pos = V[corner[:, 0]]
        uv = VT[corner[:, 1]] if len(VT) and (corner[:, 1] >= 0).all() else np.zeros((nf * 3, 2), dtype=f32)
        has_n = len(VN) and (corner[:, 2] >= 0).all()
        p3 = pos.reshape(nf, 3, 3)
        fn = np.cross(p3[:, 1] - p3[:, 0], p3[:, 2] - p3[:, 0]).astype(f32)
        if has_n:
            nrm = VN[corner[:, 2]]
        elif smooth_normals:
            acc = np.zeros((len(V), 3), dtype=np.float64)
            for c in range(3):
                np.add.at(acc, corner[c::3, 0], fn)
            nrm = acc[corner[:, 0]].astype(f32)
            nrm = nrm / np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
        else:
            fl = fn / np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-30)
            nrm = np.repeat(fl, 3, axis=0)
        nrm = nrm.astype(f32)
        # CalcTangentSpace, per face
        u3 = uv.reshape(nf, 3, 2)
        v = p3[:, 1] - p3[:, 0]
        w = p3[:, 2] - p3[:, 0]
        sx, sy = u3[:, 1, 0] - u3[:, 0, 0], u3[:, 1, 1] - u3[:, 0, 1]
        tx, ty = u3[:, 2, 0] - u3[:, 0, 0], u3[:, 2, 1] - u3[:, 0, 1]
        degenerate = (sx * ty == sy * tx)
        sx = np.where(degenerate, 0.0, sx); sy = np.where(degenerate, 1.0, sy)
        tx = np.where(degenerate, 1.0, tx); ty = np.where(degenerate, 0.0, ty)
        dirc = np.where((tx * sy - ty * sx) < 0, -1.0, 1.0)
        ftan = ((w * sy[:, None] - v * ty[:, None]) * dirc[:, None]).astype(f32)
        tan = np.repeat(ftan, 3, axis=0)
        tan = tan - nrm * np.sum(tan * nrm, axis=1, keepdims=True)
        tn = np.linalg.norm(tan, axis=1, keepdims=True)
        tan = np.where(tn > 0, tan / np.maximum(tn, 1e-30), 0.0).astype(f32)
        # JoinIdenticalVertices on (pos, normal, uv)
        key = np.concatenate([pos, nrm, uv], axis=1)
        _, first, inv = np.unique(key.view(np.uint32), axis=0, return_index=True, return_inverse=True)
        order_idx = np.argsort(first)
        remap = np.empty_like(order_idx)
        remap[order_idx] = np.arange(len(order_idx))
        inv = remap[inv.reshape(-1)]
        first = first[order_idx]
        nv = len(first)
        tacc = np.zeros((nv, 3), dtype=np.float64)
        np.add.at(tacc, inv, tan)
        tl = np.linalg.norm(tacc, axis=1, keepdims=True)
        tj = np.where(tl > 0, tacc / np.maximum(tl, 1e-30), 0.0).astype(f32)
        tri = inv.reshape(nf, 3)
        good = (tri[:, 0] != tri[:, 1]) & (tri[:, 1] != tri[:, 2]) & (tri[:, 0] != tri[:, 2])
        meshes.append(dict(material=mat, pos=pos[first], nrm=nrm[first], uv=uv[first], tan=tj,
                           faces=tri[good].astype(np.uint32)))
    return meshes, mtls
