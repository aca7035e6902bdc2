from .lss_fpn import LSSFPN

__all__ = ['LSSFPN']
