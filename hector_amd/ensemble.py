"""Synthetic perturbed-parameter ensembles: counter-based, so
member i gets the same parameters on any rank / any shard layout."""
import numpy as np

SEED = 20260928


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform01(member_index, stream, seed=SEED):
    """U[0,1) for (member, stream) pairs; member_index: int array."""
    with np.errstate(over="ignore"):
        i = np.asarray(member_index, dtype=np.uint64)
        k = _splitmix64(np.uint64(seed) + i * np.uint64(0x632BE59BD9B4E019) +
                        np.uint64(stream) * np.uint64(0xD1B54A32D192ED03))
        k = _splitmix64(k)
    return (k >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def ecs_q10(n, offset=0, seed=SEED):
    """BASELINE configs 2-4: S ~ U(1.5, 6.0) degC, q10_rh ~ U(1.0, 3.0)."""
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    S = 1.5 + 4.5 * uniform01(idx, 0, seed)
    q10 = 1.0 + 2.0 * uniform01(idx, 1, seed)
    return S, q10


def biome4(n, offset=0, seed=SEED, wf_base=1.0):
    """BASELINE config 5: 4 equal biomes, warmingfactor = wf*(1+0.5b),
    q10_rh ~ U(1,3) per biome, S ~ U(1.5,6)."""
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    S = 1.5 + 4.5 * uniform01(idx, 0, seed)
    q10 = [1.0 + 2.0 * uniform01(idx, 10 + b, seed) for b in range(4)]
    wf = [np.full(n, wf_base * (1 + 0.5 * b)) for b in range(4)]
    return S, q10, wf
