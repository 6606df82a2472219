"""doc2tex_amd: MI355X (gfx950) engine for doc2tex's recognizer forward pass.

`Model(opt)` is a drop-in for doc2tex.modules.build_model.Model; compute runs in
libd2t.so (doc2tex_amd/csrc, C-ABI in include/d2t.h).  See DESIGN.md.
"""
from .build_model import Model  # noqa: F401

__all__ = ["Model"]
