"""The host-side TCP communicator of the one-GPU shard tests.  It lives in the product now
(``vireo_amd.dist.TcpComm``, selected by ``VIREO_COMM=tcp``: ranks that share a device, which RCCL
refuses); this module keeps the tests' import path."""
from vireo_amd.dist import TcpComm          # noqa: F401
