from pyremap_amd.remapper.remapper import Remapper  # noqa: F401
