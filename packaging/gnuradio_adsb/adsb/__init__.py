"""gnuradio.adsb -- the import path gr-adsb's flowgraphs and GRC descriptors use (`import gnuradio.adsb as adsb`,
reference grc/adsb_framer.block.yml:6, grc/adsb_demod.block.yml:6) -- with the two hot-path blocks replaced by
their MI355X implementations.  Installed as <site-packages>/gnuradio/adsb/__init__.py in place of the reference's
python/adsb/__init__.py (which exports framer, demod, decoder at :24-26); see packaging/gnuradio_adsb/README.md.

    adsb.framer(fs, threshold) / .set_threshold(threshold)     -> gr_adsb_amd.blocks.framer  (reference framer.py:33-182)
    adsb.demod(fs)                                             -> gr_adsb_amd.blocks.demod   (reference demod.py:31-136)
    adsb.decoder(msg_filter, error_corr, print_level)          -> gr-adsb's own decoder.py, untouched (out of scope)

Existing .grc files, generated flowgraph scripts (examples/adsb_rx.py:182-183) and the reference's GRC descriptors
work unchanged: same ids, same constructor arguments, same ports, same tags and PDUs.
"""
from gr_adsb_amd.blocks import demod, framer  # noqa: F401

try:
    # gr-adsb's decoder.py lies next to this file in a gr-adsb install (python/adsb/decoder.py): keep using it
    from .decoder import decoder  # noqa: F401
except ImportError as _e:                       # front end installed without gr-adsb itself
    _why = str(_e)

    def decoder(*args, **kwargs):
        raise ImportError("gnuradio.adsb.decoder is gr-adsb's own decoder.py (not part of the MI355X front end): "
                          "install mhostetter/gr-adsb and copy its python/adsb/decoder.py next to %s (%s)" % (__file__, _why))
