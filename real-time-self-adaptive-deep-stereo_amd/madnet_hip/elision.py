"""Dead-store elimination over a recorded MADNet plan (post-passes of MadNetEngine.build_plan): fp32 tensors that only bf16-reading kernels consume are not
stored at all.  A mixin of MadNetEngine: the passes read the engine's tensor lists (E, Cx, dE, dCx), its library handle and its Schedule."""
import torch      # noqa: F401

from .netdef import LEVELS


class ElisionPasses(object):
    def _elide_fp32_gradient_maps(self, r):
        """Post-pass over the recorded plan (dead-store elimination): an input-gradient launch that writes the bf16 shadow of its result does not
        store the fp32 map when the ONLY op that touches that buffer afterwards is the next input gradient and that launch stages the shadow
        (mh_conv2d_takes_shadows answers for the recorded descriptor): inside the 1/4-resolution estimator and the context network the gradient
        maps then exist in bf16 only (15.7 MB less written per 128-channel layer)."""
        if not (self.sched.SHADOW_DGRAD and self.sched.SHADOW_ONLY):
            return 0
        import ctypes as C
        from . import _ffi
        ops_ = r.ops
        n = 0
        spans = []
        for idx, o in enumerate(ops_):
            if o.kind == _ffi.OP_CONV and o.i[13] == 1 and o.p[7] and not o.i[18] and o.i[22] == 1 and o.p[3]:
                spans.append((idx, int(o.p[3]), int(o.p[3]) + 4 * o.i[0] * o.i[3] * o.i[4] * o.i[16]))
        for idx, lo, hi in spans:
            if any(a < hi and lo < a + nb for _, a, nb in getattr(r, "refs", ())):
                continue                # a device table (cast / split segment) reads the map
            users = []
            for j, q in enumerate(ops_):
                if j == idx:
                    continue
                if any(q.p[k] and lo <= int(q.p[k]) < hi for k in range(len(q.p))):
                    users.append(j)
            if len(users) != 1 or users[0] < idx:
                continue
            c = ops_[users[0]]
            if not (c.kind == _ffi.OP_CONV and c.i[13] == 1 and int(c.p[0]) == lo and (c.i[23] & 1) and c.i[22] == 1):
                continue
            if sum(1 for k in range(len(c.p)) if c.p[k] and lo <= int(c.p[k]) < hi) != 1:
                continue
            if not (self._takes_shadows(c) & 1):
                continue
            ops_[idx].i[23] |= 4
            c.i[23] |= 8            # MH_CONV_IN_F32_STALE: a replay whose dispatch no longer stages the shadow is refused, not wrong
            self._note_elided(r, lo, hi - lo)
            n += 1
        return n

    def _note_elided(self, r, ptr, nbytes):
        """An fp32 buffer no op of this plan writes any more.  Kept on the recorder / plan (plan.elided) so that a reader OUTSIDE the plan can ask; with
        MH_POISON_ELIDED=1 (tests) the buffer is filled with NaN at once: an op that still reads it -- a device-table op whose table building forgot
        Recorder.note_refs, a debug read of engine.E / Cx -- then fails loudly instead of consuming a stale map."""
        if not hasattr(r, "elided"):
            r.elided = []
        r.elided.append((int(ptr), int(nbytes)))
        if self.sched.POISON_ELIDED:
            t = self._tensor_by_ptr().get(int(ptr))
            if t is not None:
                t.fill_(float("nan"))

    def _tensor_by_ptr(self):
        out = {}
        for k in LEVELS:
            for t in list(self.E[k]) + list(self.dE[k]):
                out[t.data_ptr()] = t
        for t in list(self.Cx) + list(self.dCx):
            out[t.data_ptr()] = t
        return out

    def _takes_shadows(self, c):
        """mh_conv2d_takes_shadows for a recorded OP_CONV: bit 1 = the launch stages in_shadow, bit 2 = it reads the mask from mask_shadow"""
        import ctypes as C
        from . import _ffi
        d = _ffi.ConvDesc(*([c.i[k] for k in range(18)] + [c.i[18], c.f[0], c.f[1], c.i[19], c.i[20], c.i[22]]))
        return self.lib.conv2d_takes_shadows(C.byref(d), C.c_void_p(c.p[0]), C.c_void_p(c.p[1]), C.c_void_p(c.p[6]), C.c_void_p(c.p[3]), C.c_void_p(c.p[4]))

    def _standalone_activations(self):
        """{data pointer: bytes} of the activation tensors that are allocations of their own (no view of them can start in front of them): the only
        candidates for an elided fp32 store"""
        out = {}
        for k in LEVELS:
            for t in self.E[k]:
                out[t.data_ptr()] = t.numel() * 4
        for t in self.Cx:
            out[t.data_ptr()] = t.numel() * 4
        for k in LEVELS:                    # ... and the gradient maps between the input gradients of an estimator / the context network
            for t in self.dE[k]:
                out[t.data_ptr()] = t.numel() * 4
        for t in self.dCx:
            out[t.data_ptr()] = t.numel() * 4
        return out

    def _elide_fp32_activations(self, r):
        """Post-pass (dead-store elimination, forward side): a plane-writing forward layer (OP_CONV_PLANES) does not store its fp32 result when no op
        of the recorded plan reads that tensor -- the next forward layer takes the planes, the filter gradient the hi plane, the input gradient of the
        next layer the sign of the hi plane for its leaky mask (it gets MH_CONV_MASK_F32_STALE, so a replay under another dispatch fails loudly).
        Readers are found conservatively: any pointer field of any op, and any tensor a device table of an op references (Recorder.refs), that
        OVERLAPS the buffer."""
        if not (self.use_planes and self.sched.PLANES_ONLY):
            return 0
        from . import _ffi
        ops_ = r.ops
        cand = self._standalone_activations()
        n = 0
        for idx, o in enumerate(ops_):
            if o.kind == _ffi.OP_CONV_PLANES and o.p[4] and o.p[5] and o.p[6]:
                slot = 4                    # forward: fp32 result beside both planes
            elif o.kind == _ffi.OP_CONV_PLANES_BWD and o.p[3] and o.p[4]:
                slot = 3                    # input gradient: fp32 map beside its shadow
            else:
                continue
            lo = int(o.p[slot])
            if lo not in cand:
                continue
            hi = lo + cand[lo]
            if any(a < hi and lo < a + nb for _, a, nb in getattr(r, "refs", ())):
                continue
            ok, mask_users = True, []
            for j, q in enumerate(ops_):
                if j == idx:
                    continue
                hits = [k for k in range(len(q.p)) if q.p[k] and lo <= int(q.p[k]) < hi]
                if not hits:
                    continue
                # the only tolerated readers: an input gradient (tiled families) that was given this tensor as its leaky mask TOGETHER with the mask's
                # shadow and whose kernel tests the shadow -- or as its dz together with dz's shadow and whose kernel stages the shadow
                if (q.kind == _ffi.OP_CONV and q.i[13] == 1 and hits == [4] and int(q.p[4]) == lo and (q.i[23] & 2) and q.i[22] == 1
                        and (self._takes_shadows(q) & 2)):
                    mask_users.append((q, 16))          # MH_CONV_MASK_F32_STALE
                    continue
                if (q.kind == _ffi.OP_CONV and q.i[13] == 1 and hits == [0] and int(q.p[0]) == lo and (q.i[23] & 1) and q.i[22] == 1
                        and (self._takes_shadows(q) & 1)):
                    mask_users.append((q, 8))           # MH_CONV_IN_F32_STALE
                    continue
                ok = False
                break
            if not ok:
                continue
            o.p[slot] = None
            for q, bit in mask_users:
                q.i[23] |= bit
            self._note_elided(r, lo, hi - lo)
            n += 1
        return n
