def epoch_time(start_time, end_time):
    """(minutes, seconds) of an interval — mmgclip/utils/train_utils.py:1-7."""
    elapsed = end_time - start_time
    mins = int(elapsed / 60)
    return mins, int(elapsed - mins * 60)
