"""Test helpers: convert between the oracle's scene structs and the product's ArtSceneDesc (same numbers both ways)."""
import ctypes as C

import numpy as np

import orc


def desc_from_oracle(art, cs, closest=False):
    """ArtSceneDesc of an orc.CornellScene (the reference's internal scene)."""
    s = cs.scene
    spheres = [(list(s.spheres[i].pos), s.spheres[i].r, s.spheres[i].mat) for i in range(s.n_spheres)]
    lights = []
    for i in range(s.n_lights):
        l = s.lights[i]
        lights.append(dict(shape=l.shape, mat=l.mat, boxMin=list(l.boxMin), boxMax=list(l.boxMax), normal=list(l.normal),
                           center=list(l.center), radius=l.radius, intensity=list(l.intensity), surfaceArea=l.surfaceArea))
    mats = [dict(type=s.materials[i].type, light=s.materials[i].light, p=list(s.materials[i].p)) for i in range(s.n_materials)]
    meshes = []
    if s.n_meshes:
        m = cs.mesh_arrays()
        meshes.append(dict(mode=art.MESH_CLOSEST if closest else art.MESH_REFERENCE_BF, pos=m["pos"], nrm=m["nrm"], uv=m["uv"],
                           idx=m["idx"], matid=m["matid"], bbmin=m["bbmin"], bbmax=m["bbmax"]))
    cornell = dict(min=list(s.cb_min), max=list(s.cb_max), mat=list(s.cb_mat), nrm=[list(s.cb_nrm[k]) for k in range(6)])
    return art.SceneDesc(spheres, lights, mats, meshes, cornell, list(s.cam_pos), list(s.cam_matrix))


class OracleScene:
    """orc.Scene view of an art.SceneDesc (keeps the ctypes arrays alive)."""

    def __init__(self, sd):
        d = sd.desc
        self.spheres = (orc.Sphere * max(1, d.n_spheres))()
        for i in range(d.n_spheres):
            self.spheres[i].pos = d.spheres[i].pos; self.spheres[i].r = d.spheres[i].r; self.spheres[i].mat = d.spheres[i].mat
        self.lights = (orc.Light * max(1, d.n_lights))()
        for i in range(d.n_lights):
            a, b = d.lights[i], self.lights[i]
            b.shape, b.mat, b.radius, b.surfaceArea = a.shape, a.mat, a.radius, a.surfaceArea
            b.boxMin, b.boxMax, b.normal, b.center, b.intensity = a.boxMin, a.boxMax, a.normal, a.center, a.intensity
        self.materials = (orc.Material * max(1, d.n_materials))()
        for i in range(d.n_materials):
            self.materials[i].type = d.materials[i].type; self.materials[i].light = d.materials[i].light; self.materials[i].p = d.materials[i].p
        self.meshes = (orc.Mesh * max(1, d.n_meshes))()
        for i in range(d.n_meshes):
            a, b = d.meshes[i], self.meshes[i]
            b.mode, b.nverts, b.ntris = a.mode, a.nverts, a.ntris
            b.pos = C.cast(a.pos, orc.f32p); b.nrm = C.cast(a.nrm, orc.f32p); b.uv = C.cast(a.uv, orc.f32p)
            b.idx = C.cast(a.idx, orc.i32p); b.matid = C.cast(a.matid, orc.i32p)
            b.bbmin, b.bbmax = a.bbmin, a.bbmax
        s = orc.Scene()
        s.n_spheres, s.spheres = d.n_spheres, self.spheres
        s.has_cornell = d.has_cornell
        s.cb_min, s.cb_max, s.cb_mat, s.cb_nrm = d.cb_min, d.cb_max, d.cb_mat, d.cb_nrm
        s.n_lights, s.lights = d.n_lights, self.lights
        s.n_materials, s.materials = d.n_materials, self.materials
        s.n_meshes, s.meshes = d.n_meshes, self.meshes
        s.cam_pos, s.cam_matrix = d.cam_pos, d.cam_matrix
        self.scene = s
        self._keep = sd
        s._owner = self          # `conv.OracleScene(sd).scene` alone must keep the arrays alive (found by the AddressSanitizer pass, round 3)

    def attach_bvh(self, nodes, tris, width=8):
        """Let the oracle's closest-hit mesh search walk the product's exported BVH (CPU baseline timing)."""
        self._bvh = (np.ascontiguousarray(nodes, np.float32), np.ascontiguousarray(tris, np.float32))
        for i in range(self.scene.n_meshes):
            if self.meshes[i].mode == orc.MESH_CLOSEST:
                self.meshes[i].bvh_nodes = self._bvh[0].ctypes.data_as(orc.f32p)
                self.meshes[i].bvh_tris = self._bvh[1].ctypes.data_as(orc.f32p)
                self.meshes[i].bvh_width = width


def hits_to_arrays(hits):
    n = len(hits)
    t = np.array([h.t for h in hits], np.float32)
    is_hit = np.array([h.is_hit for h in hits], np.int32)
    ptype = np.array([h.prim_type for h in hits], np.int32)
    prim = np.array([h.prim_index for h in hits], np.int32)
    mat = np.array([h.mat for h in hits], np.int32)
    nrm = np.array([list(h.normal) for h in hits], np.float32).reshape(n, 3)
    return t, is_hit, ptype, prim, mat, nrm
