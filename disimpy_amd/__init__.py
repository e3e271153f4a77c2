"""disimpy_amd - MI355X-native Monte Carlo diffusion-MRI simulator.

Drop-in for the simulation path of kerkelae/disimpy: the same
``simulations.simulation()``, ``substrates.*``, ``gradients.*`` and
``utils.vec2vec_rotmat`` API, with the random-walk step, phase accumulation and
signal sum running in hand-written HIP kernels for gfx950 behind the C ABI of
``include/dsim.h`` (``disimpy_amd/csrc``).  There is no CPU fallback.
"""
from . import gradients, simulations, substrates, utils  # noqa: F401

__version__ = "0.1.0"
