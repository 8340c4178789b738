"""Import-only stub: the reference imports cv2 at module scope; any real cv2 call must be avoided by the caller."""
class error(Exception):
    pass
INTER_LANCZOS4 = 4
INTER_CUBIC = 2
ROTATE_90_COUNTERCLOCKWISE = 2
