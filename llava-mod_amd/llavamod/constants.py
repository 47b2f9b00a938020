"""Constants shared with the reference (llavamod/constants.py:6,8)."""
IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
ALIGN_VOCAB = 151936   # hard-coded logits slice of AlignTrainer.get_p/get_logp (train/align_trainer.py:473,497)
