"""bayes_drt_amd -- MI355X-native hot path of bayes-drt behind the reference's Inverter API.

The product path is the HIP library bayes_drt_amd/libbdrt.so (C ABI: include/bdrt.h).  There is no CPU
fallback: importing the compute entry points without the library raises.
"""
from ._lib import load_library, library_path  # noqa: F401

__version__ = '0.1.0'
