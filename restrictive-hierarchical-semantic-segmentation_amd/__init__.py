"""MI355X-native hot path of Restrictive Hierarchical Semantic Segmentation.

Import as ``hrseg_amd`` (alias package at the repo root).  Sub-modules mirror the
reference's layout: ``Models.models``, ``Metrics.losses``,
``Metrics.performance_metrics``, ``train``, ``tree_util``.
"""
