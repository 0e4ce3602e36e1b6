"""Sampled parameter snapshots for the hidden-256 reference fixtures (tools/gen_golden_mappo_env.py n8m64_h256,
tools/gen_golden_learner.py ... h256): at the shipped width the two networks hold ~1 M parameters, so a fixture keeps the
parameters the run STARTS from in full (once) and every later snapshot as

    <name>#val   the values at sample_indices(numel) (all of them when numel <= N_SAMPLES)            float32
    <name>#dmax  max |delta| of the WHOLE tensor against the previous snapshot                         float64
    <name>#dl2   ||delta||_2 of the whole tensor against the previous snapshot                         float64
    <name>#cmax / #cl2   the same against the initial parameters (learner fixtures: drift since iteration 0)

The index set is a pure function of numel (numpy's legacy RandomState stream, stable across numpy versions), so generator and
test agree without storing it.  Test infrastructure only: imported by tools/gen_golden_*.py and tests/."""
import numpy as np

N_SAMPLES = 4096


def sample_indices(numel):
    if numel <= N_SAMPLES:
        return np.arange(numel, dtype=np.int64)
    return np.sort(np.random.RandomState(numel % (2 ** 31)).choice(numel, N_SAMPLES, replace=False)).astype(np.int64)


def snapshot(out, prefix, state_dict_np, prev_np, init_np=None):
    """Write the sampled snapshot of `state_dict_np` (name -> ndarray) under `prefix`; deltas against prev_np (and init_np)."""
    for k, v in state_dict_np.items():
        flat = np.asarray(v).reshape(-1)
        out[prefix + k + "#val"] = flat[sample_indices(flat.size)].astype(np.float32)
        d = flat.astype(np.float64) - np.asarray(prev_np[k], np.float64).reshape(-1)
        out[prefix + k + "#dmax"] = np.array(np.abs(d).max() if d.size else 0.0)
        out[prefix + k + "#dl2"] = np.array(np.sqrt((d * d).sum()))
        if init_np is not None:
            c = flat.astype(np.float64) - np.asarray(init_np[k], np.float64).reshape(-1)
            out[prefix + k + "#cmax"] = np.array(np.abs(c).max() if c.size else 0.0)
            out[prefix + k + "#cl2"] = np.array(np.sqrt((c * c).sum()))


def check_sampled_deltas(got_sd, before_np, Z, prefix, tol, label, which="d"):
    """got_sd: name -> torch tensor (current parameters); before_np: name -> ndarray the delta is taken against (the previous
    snapshot of THIS run for which='d' free-running comparisons is the caller's business: pass what the delta refers to);
    Z: the fixture; compares, per tensor,
        max over the sampled elements |d_got - d_ref|  <=  tol * dmax_ref        (dmax of the whole reference tensor)
        | ||d_got||_2 - ||d_ref||_2 |  <=  tol * ||d_ref||_2     and     | max|d_got| - dmax_ref |  <=  tol * dmax_ref
    where the reference's sampled delta is Z[<name>#val] - before[idx].  Returns (worst, worst_name, failures)."""
    worst, worst_name, failures = 0.0, "", []
    for k, v in got_sd.items():
        key = prefix + k
        if key + "#val" not in Z.files:
            continue
        got = v.detach().double().cpu().numpy().reshape(-1)
        p0 = np.asarray(before_np[k], np.float64).reshape(-1)
        idx = sample_indices(got.size)
        d_ref_s = Z[key + "#val"].astype(np.float64) - p0[idx]
        d_got = got - p0
        dmax, dl2 = float(Z[key + "#%smax" % which]), float(Z[key + "#%sl2" % which])
        if dmax == 0.0:
            if float(np.abs(d_got).max()) != 0.0:
                failures.append("%s %s: reference did not move, this did" % (label, k))
            continue
        errs = (float(np.abs(d_got[idx] - d_ref_s).max()) / dmax,
                abs(float(np.sqrt((d_got * d_got).sum())) - dl2) / dl2,
                abs(float(np.abs(d_got).max()) - dmax) / dmax)
        err = max(errs)
        if err > worst:
            worst, worst_name = err, k
        if not err <= tol:
            failures.append("%s %s: sampled / l2 / max update errors %.3e %.3e %.3e of max|delta| = %.3e (tolerance %.1e)"
                            % (label, k, errs[0], errs[1], errs[2], dmax, tol))
    return worst, worst_name, failures
