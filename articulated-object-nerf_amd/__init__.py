"""MI355X-native NeRF volume-rendering hot path (drop-in for zubair-irshad/articulated-object-nerf's
``NeRF.forward`` / ``render_rays`` path).  Import as ``aon_amd`` through the repo-root shim."""
__version__ = "0.1.0"
