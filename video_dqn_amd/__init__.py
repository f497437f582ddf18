"""MI355X-native Q-learning update of uiuc-robovision/video-dqn (train_q_network.py's inner loop) behind the reference's
Python surface; see DESIGN.md."""
import os

# One update runs on three HIP streams of the engine's own beside the caller's; the trainer adds a copy stream and
# torch.distributed (RCCL) one more per process group.  The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues
# (default 4), and two streams that land on one queue serialise behind each other's event waits: with the gradient all-reduce
# live that cost 1 ms of a 6.4 ms update (profiles/r02j_hw_queues.txt).  Read by the runtime when it initialises, i.e. at the
# first HIP call of the process: bench.py and train_q_network.py also set it before they import torch.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
