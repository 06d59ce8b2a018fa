"""TEST-ONLY literal restatement (pure Python, dict-based like the reference) of NarrowBandMeshField:
update_band! (src/meshfield.jl:555-588), _extrapolate_to_ghost / _nearest_band_node / _axis_slope
(:494-538) and the offset ring (:513-520).  0-based indices.  Small grids only."""
import itertools

import numpy as np

R = 6  # _BAND_SEARCH_RADIUS


def ring_offsets(N):
    # vec(collect(CartesianIndices((-R:R)^N))) is column-major (first index fastest); sort! with `by` is stable
    offs = [tuple(reversed(t)) for t in itertools.product(range(-R, R + 1), repeat=N)]
    return sorted(offs, key=lambda o: sum(c * c for c in o))


class NBRef:
    def __init__(self, vals, nlayers):
        self.n = vals.shape
        self.N = vals.ndim
        self.nlayers = nlayers
        self.d = {I: float(vals[I]) for I in np.ndindex(*vals.shape)}   # seed every node ...
        self.ring = ring_offsets(self.N)
        self.update_band()                                               # ... then restrict to the band

    def inb(self, J):
        return all(0 <= J[k] < self.n[k] for k in range(self.N))

    def nearest(self, I):
        for off in self.ring:
            J = tuple(I[k] + off[k] for k in range(self.N))
            if J in self.d:
                return J
        return None

    def axis_slope(self, P, dim, phiP):
        Jp = P[:dim] + (P[dim] + 1,) + P[dim + 1:]
        if Jp in self.d:
            return self.d[Jp] - phiP
        Jm = P[:dim] + (P[dim] - 1,) + P[dim + 1:]
        if Jm in self.d:
            return phiP - self.d[Jm]
        return 0.0

    def extrapolate(self, I):
        P = self.nearest(I)
        if P is None:
            raise ValueError("too far from the band")
        phiP = self.d[P]
        val = phiP
        for dim in range(self.N):
            delta = I[dim] - P[dim]
            if delta != 0:
                val += delta * self.axis_slope(P, dim, phiP)
        sign = lambda x: (x > 0) - (x < 0)
        return val if (phiP == 0 or sign(val) == sign(phiP)) else phiP

    def get(self, I):
        return self.d[I] if I in self.d else self.extrapolate(I)

    def update_band(self):
        N, nl = self.N, self.nlayers
        corners = list(itertools.product((0, 1), repeat=N))
        box = [o for o in itertools.product(range(-nl, nl + 1), repeat=N) if sum(abs(c) for c in o) <= nl]
        grow = {tuple(c[k] + o[k] for k in range(N)) for c in corners for o in box}
        new_keys = set()
        for I in list(self.d.keys()):
            vs = []
            ok = True
            for c in corners:
                J = tuple(I[k] + c[k] for k in range(N))
                if J not in self.d or not self.inb(J):
                    ok = False
                    break
                vs.append(self.d[J])
            if not (ok and min(vs) <= 0 <= max(vs)):
                continue
            for off in grow:
                J = tuple(I[k] + off[k] for k in range(N))
                if self.inb(J):
                    new_keys.add(J)
        self.d = {J: self.get(J) for J in new_keys}

    def mask(self):
        m = np.zeros(self.n, dtype=bool)
        for I in self.d:
            m[I] = True
        return m

    def dense(self):
        v = np.full(self.n, np.nan)
        for I, x in self.d.items():
            v[I] = x
        return v
