"""The numeric constants of data/constants.py / main.py:58-62 that the fusion training path reads."""
NUM_CLASSES = 2            # data/constants.py:95 -- survival targets (overall survival, distant metastasis)
SUPER_BATCH_SIZE = 64      # main.py:403 -- gradients are accumulated until this many patients were seen
CLASSIFICATION_THRESHOLD = 0.5   # main.py:58 -- probability above which a class counts as predicted
NUM_BOOTSTRAP_ITERATIONS = 50    # main.py:61 -- resamples of the evaluated patients under --bootstrap
