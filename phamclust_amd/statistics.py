"""Summary statistics used on the hot path and by ``SymMatrix``.

Mirrors the reference's ``phamclust.statistics`` surface
(/root/reference/src/phamclust/statistics.py:4-100): same function names,
argument meaning and error behaviour.
"""


def average(values, weights=None):
    """Weighted arithmetic mean; uniform weights when ``weights`` is falsy
    (reference statistics.py:4-22, including the float()/sum() evaluation order)."""
    if not weights:
        weights = [1] * len(values)
    if len(values) != len(weights):
        raise ValueError(f"got {len(values)} values and {len(weights)} weights")
    total = sum([v * w for v, w in zip(values, weights)])
    return float(total) / sum(weights)


def variance(values, mean=None, sample=True):
    """Sample (default) or population variance (reference statistics.py:25-51)."""
    if len(values) == 1:
        return 0.0
    if not mean:
        mean = average(values)
    spread = sum([(v - mean) ** 2 for v in values])
    return spread / (len(values) - 1 if sample else len(values))


def standard_deviation(values, mean=None, sample=True):
    """Square root of :func:`variance` (reference statistics.py:54-68)."""
    return variance(values, mean, sample) ** 0.5


def skewness(values, mean=None, sample=True):
    """Third standardised moment (reference statistics.py:71-100)."""
    if len(values) == 1:
        return 0.0
    if not mean:
        mean = average(values)
    cubes = sum([(v - mean) ** 3 for v in values])
    scale = standard_deviation(values, mean, sample) ** 3
    scale *= (len(values) - 1) if sample else len(values)
    return cubes / scale
