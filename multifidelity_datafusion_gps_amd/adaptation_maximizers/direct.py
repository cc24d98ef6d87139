"""DIRECT (DIviding RECTangles, Jones/Perttunen/Stuckman 1993) global minimiser, written for the GPU path:
every iteration evaluates ALL new sample points of ALL potentially-optimal rectangles in ONE batched
call, i.e. one predictive panel K(X*, X) on the device instead of thousands of N* = 1 callbacks.

The reference drives the Fortran DIRECT through the `DIRECT` (solve(..., maxT=50, algmethod=1),
/root/reference/src/adaptation_maximizers/DIRECT1_maximizer.py:15-26) and `scipydirect`
(minimize(func, bounds) with library defaults, scipydirect_wrapper.py:26) wrappers, neither of which is
available offline; this restates the published algorithm with the same controls:
eps (Jones' epsilon, default 1e-4), maxf, maxT, algmethod (0 = original DIRECT: size = centre-to-vertex
distance, every tied rectangle is divided; 1 = DIRECT-L, Gablonsky's locally-biased form: size = longest
side, one rectangle per size class).
"""
import numpy as np


def _potentially_optimal(sizes, fvals, fmin, eps, one_per_class):
    """indices of the potentially optimal rectangles (lower-right convex hull in the (size, f) plane + eps test)"""
    # best f per distinct size
    order = np.lexsort((fvals, sizes))
    s_sorted, f_sorted = sizes[order], fvals[order]
    first = np.r_[True, s_sorted[1:] != s_sorted[:-1]]
    cls_idx = np.flatnonzero(first)
    cs, cf = s_sorted[cls_idx], f_sorted[cls_idx]          # ascending size, min f of the class
    # lower convex hull scanning from the largest size down (only classes that can be optimal for some K > 0)
    hull = []
    for j in range(len(cs) - 1, -1, -1):
        while len(hull) >= 2:
            a, b = hull[-2], hull[-1]
            # b is above the segment a--j  -> drop it
            if (cf[b] - cf[a]) * (cs[j] - cs[a]) <= (cf[j] - cf[a]) * (cs[b] - cs[a]):
                hull.pop()
            else:
                break
        if hull and cf[j] >= cf[hull[-1]]:
            # a smaller rectangle must be strictly better than every larger hull point to matter
            continue
        hull.append(j)
    # epsilon condition against the slope to the next larger hull point
    keep = []
    for pos, j in enumerate(hull):
        if pos == 0:
            keep.append(j)  # the largest class is always potentially optimal
            continue
        big = hull[pos - 1]
        K = (cf[big] - cf[j]) / (cs[big] - cs[j])
        if cf[j] - K * cs[j] <= fmin - eps * abs(fmin) + 1e-300:
            keep.append(j)
    chosen = []
    for j in keep:
        lo = cls_idx[j]
        hi = cls_idx[j + 1] if j + 1 < len(cls_idx) else len(order)
        members = order[lo:hi]
        best = members[f_sorted[lo:hi] <= cf[j] + 1e-12 * max(1.0, abs(cf[j]))]
        chosen.extend(best[:1] if one_per_class else best)
    return chosen


def direct_minimize(f_batch, lower, upper, eps=1e-4, maxf=20000, maxT=6000, algmethod=0, fglobal=-1e100, fglper=0.01):
    """Minimise f over the box [lower, upper].  f_batch maps (B, d) -> (B,).  Returns (x, fx, info)."""
    lower = np.asarray(lower, dtype=np.float64).reshape(-1)
    upper = np.asarray(upper, dtype=np.float64).reshape(-1)
    d = lower.size
    span = upper - lower

    def evaluate(C):
        return np.asarray(f_batch(lower + C * span), dtype=np.float64).reshape(-1)

    centers = np.full((1, d), 0.5)
    levels = np.zeros((1, d), dtype=np.int64)     # number of trisections per dimension
    fvals = evaluate(centers)
    nf, it = 1, 0
    while it < maxT and nf < maxf:
        it += 1
        side = 3.0 ** (-levels.astype(np.float64))
        if algmethod == 1:
            sizes = 0.5 * side.max(axis=1)
        else:
            sizes = 0.5 * np.sqrt((side ** 2).sum(axis=1))
        sizes = np.round(sizes, 14)
        ibest = int(np.argmin(fvals))
        chosen = _potentially_optimal(sizes, fvals, fvals[ibest], eps, one_per_class=(algmethod == 1))
        # sample c +- delta e_i along the longest sides of every chosen rectangle: ONE batch
        pts, owner, dim_of, sign_of = [], [], [], []
        for r in chosen:
            lmin = levels[r].min()
            delta = 3.0 ** (-(lmin + 1.0))
            for i in np.flatnonzero(levels[r] == lmin):
                for sgn in (1.0, -1.0):
                    p = centers[r].copy()
                    p[i] += sgn * delta
                    pts.append(p); owner.append(r); dim_of.append(i); sign_of.append(sgn)
        if not pts:
            break
        pts = np.array(pts)
        fnew = evaluate(pts)
        nf += len(pts)
        owner, dim_of = np.array(owner), np.array(dim_of)
        new_levels = np.empty((len(pts), d), dtype=np.int64)
        for r in chosen:
            sel = np.flatnonzero(owner == r)
            dims = np.unique(dim_of[sel])
            w = np.array([fnew[sel[dim_of[sel] == i]].min() for i in dims])
            lev = levels[r].copy()
            for i in dims[np.argsort(w, kind="stable")]:   # best direction gets the largest children
                lev[i] += 1
                for k in sel[dim_of[sel] == i]:
                    new_levels[k] = lev
            levels[r] = lev
        centers = np.vstack([centers, pts])
        levels = np.vstack([levels, new_levels])
        fvals = np.concatenate([fvals, fnew])
        fmin = fvals.min()
        if fglobal > -1e99 and (fmin - fglobal) <= fglper / 100.0 * max(abs(fglobal), 1e-300):
            break
    ibest = int(np.argmin(fvals))
    x = lower + centers[ibest] * span
    return x, float(fvals[ibest]), dict(nf=nf, iterations=it, nrect=len(fvals))


def gablonsky_direct(f_batch, lower, upper, eps=1e-4, maxf=20000, maxT=6000, algmethod=0):
    """The reference's optimiser itself: Gablonsky's DIRECT / DIRECT-L as shipped in scipy.optimize.direct (the code the
    `DIRECT` and `scipydirect` packages wrap), with the wrappers' controls (no volume / side-length stop).  Calls
    f one point at a time, like the reference's callback.  Returns (x, fx, info) like direct_minimize."""
    from scipy.optimize import Bounds, direct
    lower = np.asarray(lower, dtype=np.float64).reshape(-1)
    upper = np.asarray(upper, dtype=np.float64).reshape(-1)
    res = direct(lambda x: float(np.asarray(f_batch(np.asarray(x)[None, :])).reshape(-1)[0]), Bounds(lower, upper),
                 eps=eps, maxfun=int(maxf), maxiter=int(maxT), locally_biased=(algmethod == 1), vol_tol=0.0, len_tol=1e-12)
    return np.asarray(res.x), float(res.fun), dict(nf=int(res.nfev), iterations=int(res.nit), message=str(res.message))
