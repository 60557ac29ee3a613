"""tracklab_amd -- MI355X-native (gfx950) implementation of TrackLab's per-frame tracking hot path
(detector pre/post-processing, ReID crop-out, association) behind TrackLab's own plugin API.

Registered with TrackLab through the ``tracklab_plugin`` entry point (pyproject.toml); Hydra finds the
yaml files of ``tracklab_amd.configs`` via ``config_package``
(hydra_plugins/tracklab_searchpath_plugin/tracklab_searchpath_plugin.py:11-20), so
``modules/track=hip_oc_sort`` etc. drop into an unmodified ``tracklab`` run.
"""
config_package = "pkg://tracklab_amd.configs"

__all__ = ["config_package"]
__version__ = "0.1.0"
