from .ClassifierExperiment import ClassifierExperiment as classification  # noqa: F401


def create_experiment(experiment_name):
    """name -> class lookup with the reference's error (mmgclip/experiments/experiments_controller.py:3-22)."""
    network_class = globals().get(experiment_name, None)
    if network_class is None:
        raise ValueError(f"Invalid network_name: {experiment_name}")
    return network_class
