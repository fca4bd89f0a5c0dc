"""The numeric constants of data/constants.py that the fusion training path reads."""
NUM_CLASSES = 2            # data/constants.py:95 -- survival targets (overall survival, distant metastasis)
SUPER_BATCH_SIZE = 64      # main.py:403 -- gradients are accumulated until this many patients were seen
