"""``rl4rs.utils.fileutil`` of the reference, served by ``rl4rs_amd.utils.fileutil``."""
from rl4rs_amd.utils.fileutil import find_match_files, find_newest_files  # noqa: F401
