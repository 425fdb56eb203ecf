from .projection import LinearProjectionLayer, MLPProjectionHead, MultiLinearHead  # noqa: F401


def get_projection_head(projection_name):
    """name -> class lookup with the reference's error (mmgclip/networks/projection_controller.py:3-23)."""
    network_class = globals().get(projection_name, None)
    if network_class is None:
        raise ValueError(f"Invalid network_name: {projection_name}")
    return network_class
