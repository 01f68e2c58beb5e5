"""gr_adsb_amd -- MI355X (gfx950) framer + demod for gr-adsb, behind the reference's block API.

    from gr_adsb_amd import framer, demod       # drop-in for `gnuradio.adsb.framer/demod`
    from gr_adsb_amd import FrontEnd            # whole-buffer / sharded front end (bench path)

The HIP library must be built first (`python -m gr_adsb_amd.build`); there is no CPU fallback.
(The directory is `gr_adsb_amd`, not `gr-adsb_amd`: a hyphen cannot be imported.)
"""
__all__ = ["framer", "demod", "FrontEnd", "modulator"]


def __getattr__(name):
    import importlib
    if name in ("framer", "demod"):
        return getattr(importlib.import_module(__name__ + ".blocks"), name)
    if name == "FrontEnd":
        return importlib.import_module(__name__ + ".frontend").FrontEnd
    if name in ("modulator", "blocks", "frontend", "grshim", "_native", "build"):
        return importlib.import_module(__name__ + "." + name)
    raise AttributeError(name)
