import sys; sys.path.insert(0, "/root/repo")
from llm_quest_amd import kernels as K
print("32x32x16", K.mfma_pipe_rate(2.0)); print("16x16x32", K.mfma_pipe_rate(2.0, shape="16x16x32"))
