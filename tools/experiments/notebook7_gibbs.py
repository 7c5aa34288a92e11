"""The reference's `7_application_dirichlet_mixture_model.ipynb` (c6-c12) on this build, as the notebook writes it: ONE trace
of `generate_data` (a `repeat` of cluster means, an inlined Dirichlet, `categorical(log probs, sample_shape=n)` and
`normal(clusters[idx], sigma)` over all n datapoints), `importance` under the data, then N_ITER Gibbs sweeps whose three
moves draw with the library and write back with `trace.update` (update_cluster_means / update_datapoint_assignment /
update_cluster_weights).  The array code between the GFI calls is torch where the notebook has jax.numpy.

  python tools/experiments/notebook7_gibbs.py [N_DATAPOINTS] [N_CLUSTERS] [N_ITER]     -> one JSON line"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import genjax_amd as genjax
from genjax_amd import ChoiceMapBuilder as C, Diff, categorical, dirichlet, gen, normal, numpy as jnp
from genjax_amd.core.pytree import Const

PRIOR_VARIANCE, OBS_VARIANCE, PRIOR_MEAN = 10.0, 1.0, 50.0


@gen
def generate_cluster(mean, var):
    cluster_mean = normal(mean, var) @ "mean"
    return cluster_mean


@gen
def generate_cluster_weight(alphas):
    probs = dirichlet(alphas) @ "probs"
    return probs


@gen
def generate_datapoints(probs, clusters, n_datapoints):
    idx = categorical(jnp.log(probs), sample_shape=n_datapoints) @ "idx"
    obs = normal(clusters[idx], OBS_VARIANCE) @ "obs"
    return obs


@gen
def generate_data(n_clusters, n_datapoints, alpha):
    clusters = generate_cluster.repeat(n=n_clusters.unwrap())(PRIOR_MEAN, PRIOR_VARIANCE) @ "clusters"
    probs = generate_cluster_weight.inline(alpha / n_clusters.unwrap() * jnp.ones(n_clusters.unwrap()))
    datapoints = generate_datapoints(probs, clusters, n_datapoints) @ "datapoints"
    return datapoints


def update_cluster_means(key, trace, n_clusters):
    ch = trace.get_choices()
    idx, x, current = ch["datapoints", "idx"].long(), ch["datapoints", "obs"], ch["clusters", "mean"]
    counts = torch.bincount(idx, minlength=n_clusters).float()
    sums = torch.zeros(n_clusters, device=x.device).index_add_(0, idx, x)
    cluster_means = sums / counts
    post_mean = PRIOR_VARIANCE / (PRIOR_VARIANCE + OBS_VARIANCE / counts) * cluster_means \
        + (OBS_VARIANCE / counts) / (PRIOR_VARIANCE + OBS_VARIANCE / counts) * PRIOR_MEAN
    post_var = 1.0 / (1.0 / PRIOR_VARIANCE + counts / OBS_VARIANCE)
    ok = counts > 0
    key, subkey = genjax.split(key)
    new_means = generate_cluster.vmap().simulate(key, (torch.where(ok, post_mean, current), torch.where(ok, post_var, torch.ones_like(post_var)))
                                                 ).get_choices()["mean"].reshape(-1)
    chosen = torch.where(ok, new_means, current)
    new_trace, _, _, _ = trace.update(subkey, C["clusters", "mean"].set(chosen), Diff.no_change(trace.get_args()))
    return new_trace


def update_datapoint_assignment(key, trace, n_clusters):
    ch = trace.get_choices()
    x, means, probs = ch["datapoints", "obs"], ch["clusters", "mean"], ch["probs"]
    # log P(idx = k) + log N(x_i; mean_k, sigma): the local densities of every (datapoint, cluster) pair
    local = torch.log(probs)[None, :] - 0.5 * ((x[:, None] - means[None, :]) / OBS_VARIANCE) ** 2
    key, subkey = genjax.split(key)
    new_idx = categorical.simulate(key, (local,)).get_choices().get_value()
    new_trace, _, _, _ = trace.update(subkey, C["datapoints", "idx"].set(new_idx), Diff.no_change(trace.get_args()))
    return new_trace


def update_cluster_weights(key, trace, n_clusters, alpha):
    counts = torch.bincount(trace.get_choices()["datapoints", "idx"].long(), minlength=n_clusters).float()
    new_alpha = alpha / n_clusters * torch.ones(n_clusters, device=counts.device) + counts
    key, subkey = genjax.split(key)
    new_probs = generate_cluster_weight.simulate(key, (new_alpha,)).get_retval()
    new_trace, _, _, _ = trace.update(subkey, C["probs"].set(new_probs), Diff.no_change(trace.get_args()))
    return new_trace


def infer(n_datapoints, n_clusters, n_iter, device):
    alpha = float(n_datapoints / (n_clusters * 10))
    per = n_datapoints // n_clusters
    offsets = PRIOR_VARIANCE * (-4 + 8 * torch.arange(n_clusters) / n_clusters)
    gen_ = torch.Generator().manual_seed(0)
    pts = (torch.rand((n_clusters, per), generator=gen_) + (PRIOR_MEAN + offsets[:, None])).reshape(-1).to(device)
    n = pts.numel()
    args = (Const(n_clusters), Const(n), alpha)
    key = genjax.key(32421)
    key, subkey = genjax.split(key)
    constraints = C["datapoints", "obs"].set(pts) | C["probs"].set(torch.ones(n_clusters, device=device) / n_clusters)
    tr, _ = generate_data.importance(subkey, constraints, args)
    score0 = float(tr.get_score())
    times = []
    for _ in range(n_iter):
        t0 = time.perf_counter()
        key, subkey = genjax.split(key)
        tr = update_cluster_means(subkey, tr, n_clusters)
        key, subkey = genjax.split(key)
        tr = update_datapoint_assignment(subkey, tr, n_clusters)
        key, subkey = genjax.split(key)
        tr = update_cluster_weights(subkey, tr, n_clusters, alpha)
        if device.type == "cuda":
            torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    return tr, (PRIOR_MEAN + offsets + 0.5).numpy(), times, score0


if __name__ == "__main__":
    n, k, it = (int(float(a)) for a in (sys.argv[1:4] + ["5000", "40", "50"][len(sys.argv) - 1:]))
    dev = genjax._lib.get().device
    tr, true_means, times, score0 = infer(n, k, it, dev)
    ch = tr.get_choices()
    means = ch["clusters", "mean"].cpu().numpy()
    counts = np.bincount(ch["datapoints", "idx"].cpu().numpy(), minlength=k)
    big = means[counts > n // (4 * k)]
    print(json.dumps({"n_datapoints": n, "n_clusters": k, "sweeps": it, "ms_per_sweep_median": 1e3 * float(np.median(times)),
                      "populated_clusters": int((counts > n // (4 * k)).sum()),
                      "max_distance_of_a_populated_cluster_to_a_true_mean": float(max(np.min(np.abs(true_means - m)) for m in big)),
                      "populated_clusters_within_1_of_a_true_mean": int(sum(np.min(np.abs(true_means - m)) < 1.0 for m in big)),
                      "score_after_importance": score0, "score": float(tr.get_score())}))
