"""Agglomerative clustering of a distance ``SymMatrix`` (the reference's clustering.py:4-51 contract).

scikit-learn's ``AgglomerativeClustering`` on the precomputed matrix does the work, as in the reference.
The hand-off is what differs: ``to_ndarray()`` is one fancy-index of the array the GPU fill produced, and the
members of each label are gathered with numpy instead of per-node Python dict traffic.
"""

import numpy as np
from sklearn.cluster import AgglomerativeClustering


def hierarchical_clustering(matrix, linkage, eps=None, n_clusters=None):
    """Sub-matrices of the clusters, largest first.  Give either ``eps`` (distance threshold) or ``n_clusters``."""
    if len(matrix) == 1:
        return [matrix]
    if not matrix.is_distance:
        raise ValueError("matrix must be a distance matrix")
    if eps is None and not n_clusters:
        raise ValueError("need either threshold or n_clusters to proceed")
    if eps and n_clusters:
        raise ValueError("threshold and n_clusters are mutually exclusive")
    labels = AgglomerativeClustering(metric="precomputed", linkage=linkage, distance_threshold=eps,
                                     n_clusters=n_clusters).fit_predict(matrix.to_ndarray())
    nodes = np.asarray(matrix.nodes, dtype=object)
    # clusters in order of first appearance of their label, as a dict keyed by label would give
    _, first = np.unique(labels, return_index=True)
    parts = [matrix.extract_submatrix(nodes[labels == labels[i]].tolist()) for i in sorted(first)]
    return sorted(parts, reverse=True)
