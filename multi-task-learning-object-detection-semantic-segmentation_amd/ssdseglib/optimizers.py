"""Optimizer description for `model.compile` -- stands in for `tf.keras.optimizers.Adam(learning_rate=1e-4)` of
NB03#cell14 (Keras 2.13 defaults: beta_1 0.9, beta_2 0.999, epsilon 1e-7).  The update itself is the fused HIP kernel
ssdseg_adam_step over the flat parameter bucket."""


class Adam:
    def __init__(self, learning_rate: float = 1e-3, beta_1: float = 0.9, beta_2: float = 0.999, epsilon: float = 1e-7):
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = float(learning_rate), float(beta_1), float(beta_2), float(epsilon)
