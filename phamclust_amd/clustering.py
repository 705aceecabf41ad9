"""Hierarchical clustering of a distance ``SymMatrix`` (reference clustering.py:4-51).

The clustering itself stays scikit-learn's ``AgglomerativeClustering`` on the precomputed
matrix, exactly as in the reference; what changes is the hand-off: ``SymMatrix.to_ndarray``
is one fancy-index of the dense array the GPU fill produced, not N^2 Python lookups.
"""

from sklearn.cluster import AgglomerativeClustering


def hierarchical_clustering(matrix, linkage, eps=None, n_clusters=None):
    """Cluster the nodes of a distance matrix; returns sub-matrices, largest first.
    ``eps`` (distance threshold) and ``n_clusters`` are mutually exclusive."""
    if len(matrix) == 1:
        return [matrix]
    if not matrix.is_distance:
        raise ValueError("matrix must be a distance matrix")
    dist = matrix.to_ndarray()
    if eps is None and not n_clusters:
        raise ValueError("need either threshold or n_clusters to proceed")
    if eps and n_clusters:
        raise ValueError("threshold and n_clusters are mutually exclusive")
    model = AgglomerativeClustering(metric="precomputed", linkage=linkage, distance_threshold=eps, n_clusters=n_clusters)
    members = dict()
    for node, label in zip(matrix.nodes, model.fit_predict(dist)):
        members.setdefault(label, set()).add(node)
    clusters = [matrix.extract_submatrix(list(nodes)) for nodes in members.values()]
    return sorted(clusters, reverse=True)
