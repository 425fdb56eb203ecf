from .losses import CLIPLoss, AveragedMedicalCLIPLoss, MMGCLIPLoss  # noqa: F401


def create_loss(loss_name):
    """name -> class lookup with the reference's error (mmgclip/loss/loss_controller.py:3-22)."""
    network_class = globals().get(loss_name, None)
    if network_class is None:
        raise ValueError(f"Invalid network_name: {loss_name}")
    return network_class
