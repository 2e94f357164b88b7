"""revisit-anything_amd: MI355X-native SegVLAD retrieval hot path (segment-VLAD -> PCA -> exact
segment kNN -> similarity-weighted image vote) behind the reference's func_vpr / place_rec_main
call surface.  See DESIGN.md.  The compute lives in csrc/ (HIP, gfx950) behind include/segvlad.h."""
__version__ = "0.1.0"
