"""Python face of the C++ host layer: the reference's scheme API (rabe::schemes::{ac17,bsw,lsw,aw11,ghw11,bdabe,mke08}),
same function names and argument order, with the Host (GPU context + randomness) as first argument.
Everything here is ctypes plumbing over include/rabe_host.h."""
from . import ac17, aw11, bdabe, bsw, ghw11, lsw, mke08  # noqa: F401
from ..hostlib import HUMAN_POLICY, JSON_POLICY, Host, RabeError, RabePanic  # noqa: F401
