"""free-hunch on MI355X: the Free Hunch guided-sampling hot path (Heun/Euler loop -> OpenAI UNet fwd+VJP ->
DCT-basis low-rank covariance with online time/space updates -> CG solve through the measurement operator)
as hand-written gfx950 kernels behind a C ABI (``include/fh_hip.h``), with the reference's Python plugin API
on top.  See DESIGN.md."""
__version__ = "0.1.0"
