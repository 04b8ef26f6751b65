"""Confidence matrix -> article labels (host side of the GNN path)."""
from .dbscan import DBScanRelation  # noqa: F401
from .textblock_clustering import TextblockClustering  # noqa: F401
