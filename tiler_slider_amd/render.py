"""Plain-text board rendering.

ref: explainrl/environment/display.py:37-79 (TextRender.render).  One character per cell with
the reference's precedence: target letter ('A' + index, or 'A' when not multi_color) over
tile letter ('a' + index) over obstacle 'X' over '.', so a tile standing on a target prints
as the target.  With show_info the three header lines "Step: i/max", "Done: bool", "" come
first.  Works on any object with the reference environment's attributes.
"""


class TextRender:
    def __init__(self, env):
        self.env = env

    def render(self, show_info=True):
        env = self.env
        state = env.state
        if state is None:
            return "Environment not initialized. Call reset() first."
        lines = [f"Step: {env.step_count}/{env.max_steps}", f"Done: {env.done}", ""] if show_info else []
        targets = [tuple(t) for t in state.target_locations]
        tiles = [tuple(t) for t in state.current_locations]
        for r in range(env.size):
            row = []
            for c in range(env.size):
                if (r, c) in targets:
                    row.append(chr(ord("A") + (targets.index((r, c)) if env.multi_color else 0)))
                elif (r, c) in tiles:
                    row.append(chr(ord("a") + (tiles.index((r, c)) if env.multi_color else 0)))
                else:
                    row.append("X" if state.is_blocked[r, c] else ".")
            lines.append("".join(row))
        return "\n".join(lines)

    __str__ = render

    @classmethod
    def simulate(cls, env, moves, print_each_step=True):
        """ref: display.py:86-125 — reset, play the moves, stop at done; True iff solved."""
        view = cls(env)
        env.reset()
        if print_each_step:
            print("Initial state:\n" + view.render() + "\n")
        for i, move in enumerate(moves):
            _, done, info = env.step(move)
            if print_each_step:
                print(f"Move {i + 1}: {move.name}\n{view.render()}\n")
            if done:
                if info.get("is_won"):
                    if print_each_step:
                        print("Puzzle solved!")
                    return True
                if info.get("timeout"):
                    if print_each_step:
                        print("Timeout!")
                    return False
        return env.state.is_won()
